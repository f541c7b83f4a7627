"""Same-box A/B of the decode GEMM shapes (M=256): average kernel time over a rotating set of weights.
    python tools/gemm_ab.py            (run once per environment setting, e.g. DIMX_GEMM_ORDER=0 / 1)"""
import os
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")
M = 256
res = []
for name, N, K, act, odt, S in (("qkv", 3456, 1152, 0, L.F32, 0), ("ff1", 4608, 1152, 3, L.BF16, 0),
                                ("ff2 (4 slabs)", 1152, 4608, 0, L.F32, 4), ("cross-q", 1152, 1152, 0, L.F32, 0)):
    a = torch.randn(M, K, device=dev).bfloat16()
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(12)]
    TILED = os.environ.get("GEMM_AB_TILED") == "1"
    if TILED:
        from dimx.engine import tile_weight
        ws = [tile_weight(w, 64) for w in ws]
    bias = torch.randn(N, device=dev)
    out = torch.empty(max(S, 1) * M, N, device=dev, dtype=torch.bfloat16 if odt == L.BF16 else torch.float32)
    flags = ((5 | (S << 16)) if S else 0) | (8 if TILED else 0)

    def run(i):
        L.check(lib.dimx_op_gemm(L.BF16, odt, L.ptr(a), K, L.ptr(ws[i % 12]), K, L.ptr(out), N, M, N, K, L.ptr(None if S else bias), act,
                                 None, N, 0, None, flags, L.stream_ptr(dev)), "gemm")
    for i in range(24):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        e0.record()
        for i in range(240):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 240 * 1e3)
    res.append("%s %.2f us" % (name, best))
# ff1 with the deferred-LayerNorm epilogue (GemmArgs.ln_stats) against the plain ff1 above
N, K = 4608, 1152
a = torch.randn(M, K, device=dev).bfloat16()
ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(12)]
bias = torch.randn(N, device=dev)
cs = torch.randn(N, device=dev)
stats = torch.rand(8, 32, 32, 2, device=dev) + 1.0
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)


def run_ln(i):
    L.check(lib.dimx_op_gemm_ln(L.BF16, L.ptr(a), L.ptr(ws[i % 12]), L.ptr(out), M, N, K, L.ptr(bias), 3, L.ptr(stats), L.ptr(cs),
                                L.stream_ptr(dev)), "gemm_ln")


if os.environ.get("DIMX_GEMM_CFG_SMALL", "34") in ("34", "35", "36", "37"):   # the ln epilogue lives in the loader/consumer kernel
    for i in range(24):
        run_ln(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0.record()
        for i in range(240):
            run_ln(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 240 * 1e3)
    res.append("ff1 + ln epilogue %.2f us" % best)
print("tiled W=%s " % os.environ.get("GEMM_AB_TILED", "0") + "DIMX_GEMM_CFG_SMALL=%s: back-to-back launches, best of 5: %s" % (os.environ.get("DIMX_GEMM_CFG_SMALL", "default"), ", ".join(res)))
