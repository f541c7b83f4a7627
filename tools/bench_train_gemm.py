"""The GEMM shapes of the training step (B=16, T=300: 4 800 rows), f32 output, per tile configuration:
    python tools/bench_train_gemm.py [cfg ...]     (0 = the library's own choice, 14 = 128x128, 34 = 64x64 loader/consumer)
fwd / dx: M = rows, (N, K) = the Linear's (out, in) or (in, out); dW: M = out, N = in, K = rows (zero-padded to 64)."""
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")


def run(M, N, K, cfg, iters=24):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(2)]
    out = torch.empty(M, N, device=dev, dtype=torch.float32)

    def one(i):
        L.check(lib.dimx_op_gemm(L.BF16, L.F32, L.ptr(a), K, L.ptr(ws[i % 2]), K, L.ptr(out), N, M, N, K, None, 0, None, 0, 0, None,
                                 cfg << 8, L.stream_ptr(dev)), "gemm")
    for i in range(4):
        one(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        one(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


R = 4800
CASES = [("fwd qkv 1152", R, 768, 1152), ("fwd out", R, 1152, 768), ("fwd ff1", R, 4608, 1152), ("fwd ff2 / dx ff1", R, 1152, 4608),
         ("fwd enc 384", R, 768, 384), ("fwd enc out", R, 384, 768), ("fwd enc ff1", R, 1536, 384), ("fwd enc ff2", R, 384, 1536),
         ("logits", R, 512, 1152), ("dW qkv", 768, 1152, 4800), ("dW out", 1152, 768, 4800), ("dW ff1", 4608, 1152, 4800),
         ("dW ff2", 1152, 4608, 4800), ("dW enc ff1", 1536, 384, 4800), ("dW enc ff2", 384, 1536, 4800), ("dW enc qkv", 768, 384, 4800)]
cfgs = [int(c) for c in sys.argv[1:]] or [0]
for name, M, N, K in CASES:
    row = []
    for cfg in cfgs:
        try:
            us = run(M, N, K, cfg)
            row.append("cfg %2d: %7.1f us %5.1f%%" % (cfg, us, 2.0 * M * N * K / us / 1e6 / 2500 * 100))
        except Exception as e:   # a configuration that does not take the shape
            row.append("cfg %2d: %s" % (cfg, str(e)[:30]))
    print("%-18s M=%-5d N=%-5d K=%-5d  %s" % (name, M, N, K, "   ".join(row)), flush=True)
