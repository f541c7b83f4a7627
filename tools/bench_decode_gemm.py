"""Decode-step GEMM shapes (M = 256) x tile configs, GPU-side durations from rocprofv3:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o g -- python tools/bench_decode_gemm.py PLAN.json
    python tools/bench_gemm.py --parse OUT PLAN.json
"""
import json
import math
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import engine as E

dev = "cuda:0"
plan = []
M = 256
shapes = [("qkv N2304 K1152", 2304, 1152, True, 0), ("self_out N1152 K768", 1152, 768, True, 0),
          ("cross_q N768 K1152", 768, 1152, True, 0), ("ff1 N4608 K1152", 4608, 1152, False, 3),
          ("ff2 N1152 K4608", 1152, 4608, True, 0), ("logits N512 K1152", 512, 1152, True, 0)]
for name, N, K, slab, act in shapes:
    a = torch.randn(M, K, device=dev)
    ws = [torch.randn(N, K, device=dev) / math.sqrt(K) for _ in range(4)]
    for cfg in (3, 32, 33):
        for sp in ((2, 4) if slab else (0,)):
            warm, iters = 3, 12
            for i in range(warm + iters):
                E.op_gemm(a, ws[i % 4], None, act, bf16=True, out_bf16=not slab, cfg=cfg, slabs=sp)
            torch.cuda.synchronize()
            plan.append((name, "c%d/s%d" % (cfg, sp), warm, iters))
json.dump(plan, open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/plan.json", "w"))
