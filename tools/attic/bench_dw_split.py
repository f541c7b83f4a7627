"""Weight-gradient GEMMs of the training step (dW [out, in] = dy^T . x, contraction over the 4 800 rows of B = 16, T = 300) with the
contraction split into f32 slabs (plain stores, added in slab order by one launch at the end of the backward pass):
    python tools/bench_dw_split.py [rows=4800]
per shape: tile configuration (34 = 64 x 64 loader / consumer waves, 14 = 128 x 128) x split count -> us of the GEMM alone."""
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4800


def run(M, N, K, cfg, sp, iters=20):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(2)]
    out = torch.empty(max(sp, 1), M, N, device=dev, dtype=torch.float32)
    flags = (cfg << 8) | ((5 | (sp << 16)) if sp > 1 else 0)

    def one(i):
        L.check(lib.dimx_op_gemm(L.BF16, L.F32, L.ptr(a), K, L.ptr(ws[i % 2]), K, L.ptr(out), N, M, N, K, None, 0, None, 0, 0, None,
                                 flags, L.stream_ptr(dev)), "gemm")
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


Kp = (R + 63) // 64 * 64
CASES = [("dec qkv", 2304, 1152), ("dec out", 1152, 768), ("dec q", 768, 1152), ("dec kv", 1536, 1152), ("dec ff1", 4608, 1152), ("dec ff2", 1152, 4608),
         ("enc qkv", 2304, 384), ("enc out", 384, 768), ("enc ff1", 1536, 384), ("enc ff2", 384, 1536), ("logits / emb", 512, 1152),
         ("vq qkv", 1536, 384), ("vq conv", 384, 1920), ("enc in", 384, 56 + 8)]
for name, M, N in CASES:
    cells = []
    for cfg in (34, 14):
        for sp in (1, 2, 3, 4, 6, 8):
            try:
                us = run(M, N, Kp, cfg, sp)
                cells.append((us, cfg, sp))
            except Exception as e:
                pass
    base = [c for c in cells if c[2] == 1]
    best = min(cells)
    print("%-14s out %4d x %-4d  " % (name, M, N) + "  ".join("%d/%d:%6.1f" % (c, s, u) for u, c, s in cells) +
          "   | best cfg %d x %d: %.1f us (unsplit best %.1f; slab traffic %.1f MB)" % (best[1], best[2], best[0], min(base)[0],
                                                                                        (best[2] + 1) * M * N * 4 / 1e6 if best[2] > 1 else 0), flush=True)
