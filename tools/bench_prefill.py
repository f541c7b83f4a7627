"""Prefill GEMM shapes of the C3 workload, event-timed (kernels of 100+ us: host overhead is negligible).
    [DIMX_TILE_MAP=1] python tools/bench_prefill.py [cfg ...]"""
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")


def run(M, N, K, out_bf16, act, cfg, iters=16):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(2)]
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)

    def one(i):
        L.check(lib.dimx_op_gemm(L.BF16, L.BF16 if out_bf16 else L.F32, L.ptr(a), K, L.ptr(ws[i % 2]), K, L.ptr(out), N, M, N, K,
                                 L.ptr(bias) if act else None, act, None, 0, 0, None, cfg << 8, L.stream_ptr(dev)), "gemm")
    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        one(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


CASES = [("cross_kv", 76800, 1536, 1152, True, 0), ("vq_l1", 76800, 1536, 384, True, 2), ("vq_l2", 76800, 384, 1536, False, 0),
         ("enc_qkv", 76800, 2304, 384, True, 0), ("dec_ff1_tf", 76544, 4608, 1152, True, 3),
         ("dec_ff2_tf", 76544, 1152, 4608, False, 0), ("dec_qkv_tf", 76544, 2304, 1152, True, 0)]
cfgs = [int(c) for c in sys.argv[1:]] or [0]
for name, M, N, K, obf, act in CASES:
    row = []
    for cfg in cfgs:
        us = run(M, N, K, obf, act, cfg)
        row.append("cfg %2d: %8.1f us %5.1f%%" % (cfg, us, 2.0 * M * N * K / us / 1e6 / 2500 * 100))
    print("%-11s M=%-6d N=%-5d K=%-5d  %s" % (name, M, N, K, "   ".join(row)), flush=True)
