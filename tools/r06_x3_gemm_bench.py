"""Round 6: the split-bf16 decode GEMM (csrc/gemm_x3.hip) per decoder shape, back-to-back launch time (HIP events), next to the
exact-f32 MFMA kernel it replaces.  DIMX_X3_ABL=1 (no LDS-DMA in the loop) / 2 (no split, no MFMA) / DIMX_X3_STAGES=5: one setting per
process (they are read once)."""
import os
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

dev = torch.device("cuda:0")
lib = L.load()
SHAPES = [("qkv", 2304, 1152, True), ("self/cross out", 1152, 768, True), ("cross q", 768, 1152, True), ("ff1 (+GELU)", 4608, 1152, False),
          ("ff2", 1152, 4608, True), ("logits", 512, 1152, True)]
M = 256


def run(name, N, K, slabs, x3, iters=200):
    from dimx import engine as E
    a = torch.randn(M, K, device=dev)
    ws = [torch.randn(N, K, device=dev) / K ** 0.5 for _ in range(4)]
    planes = [E.op_split_x3(w) for w in ws] if x3 else None
    ns = lib.dimx_op_gemm_slabs(L.F32, M, N, K, (16 if x3 else 0) | 5) if slabs else 1
    out = torch.empty(ns, M, N, device=dev)
    bias = torch.zeros(N, device=dev)
    for i in range(iters):
        if x3:
            L.check(lib.dimx_op_gemm_x3(L.ptr(a), K, L.ptr(planes[i % 4]), L.ptr(out), N, M, N, K, None if slabs else L.ptr(bias),
                                        0 if slabs else 3, None, 0, 4 if slabs else 0, L.stream_ptr(dev)), "gemm_x3")
        else:
            L.check(lib.dimx_op_gemm(L.F32, L.F32, L.ptr(a), K, L.ptr(ws[i % 4]), K, L.ptr(out), N, M, N, K, None if slabs else L.ptr(bias),
                                     0 if slabs else 3, None, 0, 0, None, 5 if slabs else 0, L.stream_ptr(dev)), "gemm")
    torch.cuda.synchronize()


for name, N, K, slabs in SHAPES:
    run(name, N, K, slabs, True, iters=60)
    if not os.environ.get("DIMX_X3_ABL"):
        run(name, N, K, slabs, False, iters=60)
print("done")
