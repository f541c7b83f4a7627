"""(the cfg 73 / gemm_dec route exists at commit 044eb4c only)
In-kernel wall-clock stamps (100 MHz) of a decode GEMM block, any shape / kernel (round 5):
    DIMX_GEMM_PROF=1 python tools/r05_gemm_stamps.py CFG N K SLABS ACT
CFG 34 = 64 x 64 loader/consumer kernel, 72 = 64 x 72 one-block-per-CU kernel."""
import os
import sys

import torch

os.environ["DIMX_GEMM_PROF"] = "1"
sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

cfg, N, K, S, act = (int(x) for x in sys.argv[1:6])
FRAG = cfg == 73   # gemm_dec_kernel on fragment-packed weights
if FRAG:
    cfg = 0
lib = L.load()
dev = torch.device("cuda:0")
M = 256
a = torch.randn(M, K, device=dev).bfloat16()
ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(6)]
bias = torch.randn(N, device=dev)
obf = S == 0
out = torch.empty(max(S, 1) * M, N, device=dev, dtype=torch.bfloat16 if obf else torch.float32)
bn = 72 if (cfg == 72 or FRAG) else 64
nblk = (M // 64) * ((N + bn - 1) // bn) * max(S, 1)
SL = 64
prof = torch.zeros(nblk * SL, dtype=torch.int64, device=dev)
acc = torch.zeros(nblk, SL, dtype=torch.float64)
n = 0
flags = ((5 | (S << 16)) if S else 0) | (cfg << 8) | (16 if FRAG else 0)
if FRAG:
    from dimx import engine
    wfs = [engine.op_pack_w_frag(w) for w in ws]
for i in range(20):
    prof.zero_()
    if FRAG:
        L.check(lib.dimx_op_gemm_dec(L.ptr(a), K, L.ptr(wfs[i % 6]), L.ptr(out), N, L.BF16 if obf else L.F32, M, N, K,
                                     L.ptr(bias if not S else None), act, S, None, None, L.ptr(prof), L.stream_ptr(dev)), "gemm_dec")
    else:
      L.check(lib.dimx_op_gemm(L.BF16, L.BF16 if obf else L.F32, L.ptr(a), K, L.ptr(ws[i % 6]), K, L.ptr(out), N, M, N, K,
                             L.ptr(bias if not S else None), act, L.ptr(prof), N, 0, None, flags, L.stream_ptr(dev)), "gemm")
    torch.cuda.synchronize()
    st = prof.view(nblk, SL).cpu().double()
    if i >= 6:
        acc += st - st[:, :1].min()
        n += 1
acc /= n * 100.0
nk = K // 64 // max(S, 1)
c = acc
nit = min(nk, 12)


def mm(col):
    return "%.2f / %.2f" % (c[:, col].mean(), c[:, col].max())


print(("gemm_dec " if FRAG else "") + "cfg %d M=256 N=%d K=%d slabs=%d act=%d: %d blocks, %d k-tiles per block; us after the first block's start, mean / max over blocks" % (cfg, N, K, S, act, nblk, nk))
print("  consumer 0: start %s | first tile released %s | main loop done %s | bias+args %s | act %s | stores issued %s | acked %s" % (
    mm(0), mm(3), mm(28), mm(24), mm(25), mm(26), mm(29)))
if nit >= 3:
    rel = [c[:, 3 + 2 * it].mean() for it in range(nit)]
    print("  consumer 0: released-to-released per k-tile: " + " ".join("%.2f" % (rel[i + 1] - rel[i]) for i in range(nit - 1)))
    wait = [(c[:, 3 + 2 * it] - c[:, 2 + 2 * it]).mean() for it in range(nit)]
    print("  consumer 0: time at the barrier per k-tile:  " + " ".join("%.2f" % w for w in wait))
print("  loader 0:   start %s | prologue issued %s | last barrier %s" % (mm(32), mm(33), mm(32 + 28)))
if nit >= 3:
    iss = [(c[:, 32 + 2 + 2 * it]).mean() for it in range(nit)]
    land = [(c[:, 32 + 3 + 2 * it] - c[:, 32 + 2 + 2 * it]).mean() for it in range(nit)]
    print("  loader 0:   loop-top to loop-top per k-tile:  " + " ".join("%.2f" % (iss[i + 1] - iss[i]) for i in range(nit - 1)))
    print("  loader 0:   vmcnt wait per k-tile:            " + " ".join("%.2f" % w for w in land))
