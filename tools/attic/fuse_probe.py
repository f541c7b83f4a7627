"""Go / no-go probe of VERDICT round 2, item 2 (hide the decode step's latency-bound GEMM chain under its HBM-bound attention
by horizontal fusion over two half-batches): ONE launch whose blocks run either the decode cross-attention of 128 clips or the
first feed-forward projection at M = 128 (csrc/gemm.hip fused_probe_kernel, both kernel bodies are the product's), against the
same two pieces of work as the decode step launches them, back to back.  Also timed: each role alone in the fused kernel's
geometry (512-thread blocks, 64 KiB of LDS), which separates what the changed geometry costs from what co-residency gives.
Results are checked against the separate launches.  GO if fused <= 0.75 x (attention + GEMM).
    python tools/fuse_probe.py [clips=128] [keys=300]
"""
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
H, Tp = 12, (T + 7) // 8 * 8
M, N, K = Bc, 4608, 1152
LAYERS = 4

torch.manual_seed(0)
kc = [torch.randn(Bc, H, Tp, 64, device=dev).bfloat16() for _ in range(LAYERS)]
vc = [torch.randn(Bc, H, Tp, 64, device=dev).bfloat16() for _ in range(LAYERS)]
q = torch.randn(Bc, H * 64, device=dev)
km = torch.ones(Bc, T, dtype=torch.uint8, device=dev)
a = torch.randn(M, K, device=dev).bfloat16()
w = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(LAYERS)]
bias = torch.randn(N, device=dev)
o_sep = torch.empty(Bc, H * 64, device=dev, dtype=torch.bfloat16)
o_fus = torch.empty_like(o_sep)
c_sep = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
c_fus = torch.empty_like(c_sep)
st = L.stream_ptr(dev)


def attn_sep(i):
    L.check(lib.dimx_op_decode_attn(L.BF16, L.ptr(q), L.ptr(kc[i % LAYERS]), L.ptr(vc[i % LAYERS]), L.ptr(o_sep), Bc, H, Tp, T, 0.125,
                                    L.ptr(km), 0, 1, st), "decode_attn")


def gemm_sep(i):
    L.check(lib.dimx_op_gemm(L.BF16, L.BF16, L.ptr(a), K, L.ptr(w[i % LAYERS]), K, L.ptr(c_sep), N, M, N, K, L.ptr(bias), 3, None, 0,
                             0, None, 0, st), "gemm")


def fused(i, which, hw=None):
    L.check(lib.dimx_op_fused_probe(L.ptr(a), L.ptr(w[i % LAYERS]), L.ptr(bias), L.ptr(c_fus), L.BF16, M, N, K, 3, L.ptr(q),
                                    L.ptr(kc[i % LAYERS]), L.ptr(vc[i % LAYERS]), L.ptr(o_fus), Bc, H, Tp, T, 0.125, L.ptr(km), which,
                                    L.ptr(hw) if hw is not None else None, st), "fused_probe")


def timeit(fn, iters=200, warm=20):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


# correctness: the fused launch computes what the separate launches compute
attn_sep(0); gemm_sep(0); fused(0, 0)
torch.cuda.synchronize()
# (the product launch may split a (clip, head)'s keys over 2 waves at this batch size: other summation order, 1 bf16 ulp)
assert (o_sep.float() - o_fus.float()).abs().max().item() <= 4e-3, "attention role differs: %g" % (o_sep.float() - o_fus.float()).abs().max().item()
assert torch.equal(c_sep, c_fus), "GEMM role differs: %g" % (c_sep.float() - c_fus.float()).abs().max().item()
n_gemm, n_attn = (M + 63) // 64 * ((N + 63) // 64), (Bc * H + 7) // 8
grid = (n_gemm + n_attn + 15) // 16 * 16
hw = torch.zeros(grid, dtype=torch.int32, device=dev)
fused(0, 0, hw)
torch.cuda.synchronize()
hv = hw.cpu().numpy().astype("uint32")
placed = {}
for v in hv:
    if v == 0:
        continue
    role, xcc = int(v >> 28) & 1, int(v >> 24) & 7
    cu = (xcc, int(v >> 13) & 0x7, int(v >> 12) & 0x1, int(v >> 8) & 0xf)   # (XCC, SE_ID, SH_ID, CU_ID) of HW_REG_HW_ID
    placed.setdefault(cu, set()).add(role)
both = sum(1 for r in placed.values() if len(r) == 2)
print("blocks: %d GEMM + %d attention in a grid of %d; %d CUs hosted blocks, %d of them both kinds" % (n_gemm, n_attn, grid, len(placed), both))

rows = []
for rnd in range(3):
    t_a = timeit(attn_sep)
    t_g = timeit(gemm_sep)
    t_seq = timeit(lambda i: (attn_sep(i), gemm_sep(i)))
    t_f = timeit(lambda i: fused(i, 0))
    t_fa = timeit(lambda i: fused(i, 2))
    t_fg = timeit(lambda i: fused(i, 1))
    rows.append((t_a, t_g, t_seq, t_f, t_fa, t_fg))
    print("round %d: attention %.2f us, GEMM %.2f us, back to back %.2f us | fused %.2f us (attention role alone %.2f, GEMM role alone %.2f)"
          % ((rnd,) + rows[-1]), flush=True)
med = [sorted(r[i] for r in rows)[1] for i in range(6)]
ratio = med[3] / med[2]
print("median: back to back %.2f us, fused %.2f us -> fused / back-to-back = %.3f  (%s: threshold 0.75)" % (med[2], med[3], ratio, "GO" if ratio <= 0.75 else "NO-GO"))
