"""The 384-wide f32 projections of the prefill (attention out / MLP second layer of the VQ and encoder stacks: residual epilogue),
event-timed:  [DIMX_LIB=...] python tools/bench_prefill_res.py"""
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")


def run(M, N, K, with_res, iters=16):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(2)]
    res = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32)

    def one(i):
        L.check(lib.dimx_op_gemm(L.BF16, L.F32, L.ptr(a), K, L.ptr(ws[i % 2]), K, L.ptr(out), N, M, N, K, None, 0,
                                 L.ptr(res) if with_res else None, N, 0, None, 0, L.stream_ptr(dev)), "gemm")
    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        one(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    ref = (a[:4096].float() @ ws[(iters - 1) % 2].float().t()) + (res[:4096] if with_res else 0)
    err = (out[:4096] - ref).abs().max().item()
    return us, err


for M, N, K in ((76800, 384, 384), (76800, 384, 768), (76800, 384, 1536), (76544, 1152, 768), (76544, 1152, 4608)):
    u0, e0 = run(M, N, K, False)
    u1, e1 = run(M, N, K, True)
    mb = (M * K * 2 + 2 * M * N * 4) / 1e6
    print("M=%d N=%d K=%d  f32 out: %7.1f us   + residual: %7.1f us  (%.0f MB -> %.2f TB/s)   max err %.2e / %.2e" %
          (M, N, K, u0, u1, mb, mb / u1, e0, e1), flush=True)
