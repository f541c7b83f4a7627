"""Decode-step GEMM shapes (M = 256) in the f32 parity mode, event-timed through dimx_op_gemm (weights rotate over 4 buffers).
    python tools/bench_f32_decode_gemm.py"""
import math
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")
M = 256
SHAPES = [("qkv N2304 K1152", 2304, 1152, (1, 2, 3), 0), ("self_out N1152 K768", 1152, 768, (1, 2, 4), 0), ("cross_q N768 K1152", 768, 1152, (1, 3, 6), 0),
          ("ff1 N4608 K1152 (gelu, no slabs)", 4608, 1152, (0,), 3), ("ff2 N1152 K4608", 1152, 4608, (1, 2, 4, 8), 0), ("logits N512 K1152", 512, 1152, (1, 4, 6), 0)]
for name, N, K, splits, act in SHAPES:
    a = torch.randn(M, K, device=dev)
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).contiguous() for _ in range(4)]
    bias = torch.randn(N, device=dev)
    row = []
    for sp in splits:
        out = torch.empty(max(sp, 1), M, N, device=dev)
        flags = (5 if sp else 0) | (sp << 16)

        def one(i):
            L.check(lib.dimx_op_gemm(L.F32, L.F32, L.ptr(a), K, L.ptr(ws[i % 4]), K, L.ptr(out), N, M, N, K, L.ptr(bias) if act else None, act,
                                     None, 0, 0, None, flags, L.stream_ptr(dev)), "gemm")
        for i in range(4):
            one(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            one(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 40
        row.append("split %d: %6.1f us (%4.1f %% of 157 TF)" % (sp, us, 2.0 * M * N * K / us / 1e6 / 157.3 * 100))
    print("%-34s %s" % (name, "   ".join(row)), flush=True)
