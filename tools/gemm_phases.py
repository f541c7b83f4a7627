"""Where does a decode GEMM block spend its time?  In-kernel wall-clock stamps (100 MHz) on the ff1 shape (M=256, N=4608,
K=1152, GELU, bf16 out, no split-K), for the 4-wave kernel (cfg 3, ABL = 3 instantiation of gemm_glds_kernel) and for the
loader/consumer kernel (cfg 34, PROF instantiation of gemm_ws_kernel).
    DIMX_GEMM_PROF=1 python tools/gemm_phases.py [3|34]"""
import os
import sys

import torch

os.environ["DIMX_GEMM_PROF"] = "1"
sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
LN = len(sys.argv) > 2 and sys.argv[2] == "ln"   # the deferred-LayerNorm epilogue of ff1 (dimx_op_gemm_ln; cfg 34 only)
lib = L.load()
dev = torch.device("cuda:0")
M, N, K = 256, 4608, 1152
a = torch.randn(M, K, device=dev).bfloat16()
ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(6)]
bias = torch.randn(N, device=dev)
stats = torch.rand(8, 32, 32, 2, device=dev) + 1.0
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
nblk = (M // 64) * (N // 64)
SL = 32 if cfg == 3 else 64
prof = torch.zeros(nblk * SL, dtype=torch.int64, device=dev)
acc = torch.zeros(nblk, SL, dtype=torch.float64)
n = 0
for i in range(20):
    prof.zero_()
    if LN:
        L.check(lib.dimx_op_gemm_ln(L.BF16, L.ptr(a), L.ptr(ws[i % 6]), L.ptr(out), M, N, K, L.ptr(prof), 3, L.ptr(stats),
                                    L.ptr(bias), L.stream_ptr(dev)), "gemm_ln")
    else:
        L.check(lib.dimx_op_gemm(L.BF16, L.BF16, L.ptr(a), K, L.ptr(ws[i % 6]), K, L.ptr(out), N, M, N, K, L.ptr(bias), 3,
                                 L.ptr(prof), N, 0, None, cfg << 8, L.stream_ptr(dev)), "gemm")
    torch.cuda.synchronize()
    st = prof.view(nblk, SL).cpu().double()
    if i >= 6:
        acc += st - st[:, :1].min()
        n += 1
acc /= n * 100.0
print("ff1 M=256 N=4608 K=1152, 64x64 tiles, 288 blocks, cfg %d; us relative to the first block's start (mean / max over blocks)" % cfg)


def table(cols, names):
    prev = None
    for j, name in zip(cols, names):
        col = acc[:, j]
        print("  %-34s mean %6.2f  max %6.2f%s" % (name, col.mean(), col.max(), "" if prev is None else "  (+%.2f)" % (col.mean() - prev)))
        prev = col.mean()


if cfg == 3:
    inner = acc[:, 20:24] - acc[:, 10:11]          # iteration 8 begins at stamp 10
    print("  inside iteration 8 (us after its start, wave 0): vmcnt wait done %.3f, barrier passed %.3f, DMAs issued %.3f, "
          "reads + MFMAs issued %.3f" % tuple(inner.mean(0).tolist()))
    table(list(range(20)) + [28, 29], ["start", "prologue DMAs issued"] + ["iteration %d begins" % i for i in range(18)]
          + ["main loop done", "epilogue stored"])
else:
    print(" consumer wave 0:")
    names = ["start"]
    cols = [0]
    for it in range(12):
        cols += [2 + 2 * it, 3 + 2 * it]
        names += ["it %d: at the barrier" % it, "it %d: released" % it]
    table(cols + [28, 24, 25, 26, 29], names + ["main loop done", "epilogue: arguments + bias loaded", "epilogue: activation done",
                                                "epilogue: stores issued", "epilogue: stores acknowledged"])
    print(" loader wave 4:")
    names = ["start", "prologue DMAs issued"]
    cols = [32, 33]
    for it in range(12):
        cols += [32 + 2 + 2 * it, 32 + 3 + 2 * it]
        names += ["it %d: issue done, waiting" % it, "it %d: own pieces landed" % it]
    table(cols + [32 + 28], names + ["last barrier passed"])
