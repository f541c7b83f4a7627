"""Throughput of the widened rows at the C3 shape, bf16 perf mode, synthetic weights/clips.
    python tools/bench_variants.py [--batch 256] [--frames 300] [--steps 3]
"""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L, prng
from dimx.seq2seq import ListenerGenerator
from dimx.seq2seq_pretrain import SLM

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
B, T = a.batch, a.frames
dev = torch.device("cuda:0")
v_s824 = torch.from_numpy(prng.normal(1, "bv.s", (B, T, 824))).to(dev)
v_l = torch.from_numpy(prng.normal(1, "bv.l", (B, T, 56))).to(dev)
v_a = v_s824[..., 56:].contiguous()
v_s = v_s824[..., :56].contiguous()
mask = torch.ones(B, T, dtype=torch.bool, device=dev)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / a.steps


m = ListenerGenerator(numeric_mode=L.MODE_PERF_BF16).to(dev)
t = timed(lambda: m.generate(v_s824, v_l, mask, seed=7))
print("legacy ListenerGenerator.generate  B=%d T=%d: %.1f ms/batch  %.0f clips/s" % (B, T, t * 1e3, B / t))
t = timed(lambda: m(v_s824, v_l, mask))
print("legacy ListenerGenerator.forward   B=%d T=%d: %.1f ms/batch  %.0f clips/s" % (B, T, t * 1e3, B / t))
del m
s = SLM(numeric_mode=L.MODE_PERF_BF16).to(dev)
t = timed(lambda: s(v_s, v_l, v_a, mask))
print("SLM.forward (pre-training fwd)     B=%d T=%d: %.1f ms/batch  %.0f clips/s" % (B, T, t * 1e3, B / t))
