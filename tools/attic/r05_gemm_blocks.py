"""Round 5, experiment 1: how much of a decode GEMM's time is the 288-blocks-on-256-CUs tail?
The three chip-wide decode GEMMs (M = 256) launch 288 blocks of 64 x 64 tiles: 32 CUs host two blocks.  Same kernels at
N chosen so that the launch is 252 / 256 / 288 / 320 / 512 blocks; GPU-side durations from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d OUT -o g -- python tools/r05_gemm_blocks.py PLAN.json
    python tools/bench_gemm.py --parse OUT PLAN.json
"""
import json
import math
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")
M = 256
plan = []


def run(label, N, K, act, out_bf16, slabs, bias_on, iters=24, ncopies=8, cfg=0, frag=False):
    a = torch.randn(M, K, device=dev).bfloat16()
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(ncopies)]
    bias = torch.randn(N, device=dev) if bias_on else None
    out = torch.empty(max(slabs, 1) * M, N, device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    flags = ((5 | (slabs << 16)) if slabs else 0) | (cfg << 8)
    warm = 4
    for i in range(warm + iters):
        L.check(lib.dimx_op_gemm(L.BF16, L.BF16 if out_bf16 else L.F32, L.ptr(a), K, L.ptr(ws[i % ncopies]), K, L.ptr(out), N, M, N, K,
                                 L.ptr(bias), act, None, N, 0, None, flags, L.stream_ptr(dev)), "gemm")
    torch.cuda.synchronize()
    tiles = 4 * ((N + 63) // 64) * max(slabs, 1)
    plan.append((label, "N%d/s%d/c%d%s" % (N, slabs, cfg, "/frag" if frag else ""), warm, iters))


if len(sys.argv) > 2 and sys.argv[2] == "ws72":   # the real shapes: 64 x 64 tiles (cfg 34, 288 blocks) vs 64 x 72 (cfg 72, 256 blocks)
    for cfg, fr in ((34, False), (72, False)):
        run("ff1 N4608 K1152 gelu bf16-out", 4608, 1152, 3, True, 0, True, cfg=cfg, frag=fr)
    for cfg, fr in ((34, False), (72, False)):
        run("ff2 N1152 K4608 4 slabs", 1152, 4608, 0, False, 4, False, cfg=cfg, frag=fr)
    for cfg, fr in ((34, False), (72, False)):
        run("qkv N2304 K1152 2 slabs", 2304, 1152, 0, False, 2, False, cfg=cfg, frag=fr)
    for cfg, fr in ((34, False), (72, False)):
        run("self-out N1152 K768 2 slabs", 1152, 768, 0, False, 2, False, cfg=cfg, frag=fr)
    json.dump(plan, open(sys.argv[1], "w"))
    sys.exit(0)
for N in (4032, 4096, 4608, 5120, 8192):
    run("ff1-like K1152 gelu bf16-out", N, 1152, 3, True, 0, True)
for N in (4032, 4096, 4608):
    run("ff1-like K1152 no-bias no-act", N, 1152, 0, True, 0, False)
for N in (1024, 1152, 1280):
    run("ff2-like K4608 4 slabs", N, 4608, 0, False, 4, False)
for N in (1152,):
    for s in (2, 3, 4, 6, 8):
        run("ff2 K4608 N1152 by slabs", N, 4608, 0, False, s, False)
for N in (2048, 2304, 2560):
    run("qkv-like K1152 2 slabs", N, 1152, 0, False, 2, False)
for s in (1, 2, 3, 4):
    run("qkv K1152 N2304 by slabs", 2304, 1152, 0, False, s, False)
json.dump(plan, open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/plan.json", "w"))
