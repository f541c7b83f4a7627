"""GEMM microbenchmark on the GPU box: every dense shape of the C3 workload x tile/stage configs.
    python tools/bench_gemm.py [decode|prefill|all]
"""
import math
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

dev = torch.device("cuda:0")
lib = L.load()


def run(M, N, K, out_bf16, inplace, cfg, splitk, act=0, iters=30, ncopies=6):
    a = (torch.randn(M, K, device=dev)).bfloat16()
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(ncopies)]
    res = torch.randn(M, N, device=dev) if inplace else None
    out = res if inplace else torch.empty(M, N, device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    flags = (1 if inplace else 0) | (cfg << 8) | (splitk << 16)

    def f(i):
        L.check(lib.dimx_op_gemm(L.BF16, L.BF16 if out_bf16 else L.F32, L.ptr(a), K, L.ptr(ws[i % ncopies]), K,
                                 L.ptr(out), N, M, N, K, None, act, L.ptr(res), N if inplace else 0, 0, None, flags,
                                 L.stream_ptr(dev)), "gemm")
    for i in range(5):
        f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        f(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us / 1e6


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("decode", "all"):
    print("== decode step GEMMs (M=256), us [TFLOP/s] per config ==")
    shapes = [("qkv", 2304, 1152, True, False, 0), ("self_out", 1152, 768, False, True, 0),
              ("cross_q", 768, 1152, True, False, 0), ("ff1", 4608, 1152, True, False, 3),
              ("ff2", 1152, 4608, False, True, 0), ("logits", 512, 1152, False, False, 0)]
    for name, N, K, obf, inpl, act in shapes:
        row = []
        for cfg in (3, 4, 7, 1):
            for sp in ((0, 2, 4, 8) if inpl else (0,)):
                us, tf = run(256, N, K, obf, inpl, cfg, sp, act)
                row.append("c%d/s%d:%.1f" % (cfg, sp, us))
        print("%-9s N=%4d K=%4d  " % (name, N, K) + "  ".join(row))
if which in ("prefill", "all"):
    print("== prefill GEMMs (M=76800) ==")
    shapes = [("vq_qk", 768, 384, True, False, 0), ("vq_out", 384, 384, False, True, 0), ("vq_l1", 1536, 384, True, False, 2),
              ("vq_l2", 384, 1536, False, True, 0), ("xe_qkv", 2304, 384, True, False, 0), ("xe_out", 384, 768, False, True, 0),
              ("ckv", 1536, 1152, True, False, 0)]
    for name, N, K, obf, inpl, act in shapes:
        row = []
        for cfg in (1, 2, 3, 4, 6, 7, 8):
            us, tf = run(76800, N, K, obf, inpl, cfg, 0, act, iters=8, ncopies=2)
            row.append("c%d:%.0fus/%.0fTF" % (cfg, us, tf))
        print("%-7s N=%4d K=%4d  " % (name, N, K) + "  ".join(row))
