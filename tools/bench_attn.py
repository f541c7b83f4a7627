"""Prefill attention at the C3 shapes, event-timed: the row-major-V kernel of attention_tr.hip (default) or, with
DIMX_ATTN_OLD=1, round 3's attn_kernel<bf16, 2, 64, VROW>.
    [DIMX_ATTN_OLD=1] python tools/bench_attn.py [B]"""
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def run(H, Lq, Lk, D, causal, kmask, iters=12, nbuf=3):
    C = H * D
    bufs = [tuple(torch.randn(B, L_, C, device=dev).to(torch.bfloat16) for L_ in (Lq, Lk, Lk)) for _ in range(nbuf)]
    out = torch.empty(B, Lq, C, device=dev, dtype=torch.bfloat16)
    km = torch.ones(B, Lk, dtype=torch.uint8, device=dev) if kmask else None
    scale = 384 ** -0.5 if D == 48 else 0.125

    def one(i):
        q, k, v = bufs[i % nbuf]
        L.check(lib.dimx_op_attention_rowv(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), B, H, Lq, Lk, D, C, C, C, C, scale,
                                           1 if causal else 0, None, L.ptr(km), L.stream_ptr(dev)), "attention_rowv")
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
    flops = 4.0 * B * H * Lq * Lk * D * (0.5 if causal else 1.0)
    byts = 2.0 * B * C * (2 * Lq + 2 * Lk)
    return us, flops / us / 1e6, byts / us / 1e3


CASES = [("vq (8 x 48, clip lengths)", 8, 300, 300, 48, False, False), ("vq decode (8 x 48, 299)", 8, 299, 299, 48, False, False),
         ("x-enc causal + mask (12 x 64)", 12, 300, 300, 64, True, True), ("dec-tf cross (12 x 64, 299 x 300)", 12, 299, 300, 64, False, True),
         ("non-causal 12 x 64", 12, 300, 300, 64, False, False), ("long 12 x 64 T 1500 causal", 12, 1500, 1500, 64, True, False)]
for name, H, Lq, Lk, D, causal, km in CASES:
    if Lq > 1000 and B > 64:
        continue
    us, tf, gbs = run(H, Lq, Lk, D, causal, km)
    print("%-36s %8.1f us  %7.1f TFLOP/s (%4.1f %% of 2500; causal counts half)  %6.0f GB/s of q+k+v+o" % (name, us, tf, tf / 25.0, gbs), flush=True)
