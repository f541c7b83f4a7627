"""Launch one GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py M N K cfg out_bf16 act"""
import math
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

M, N, K, cfg, obf, act = [int(v) for v in sys.argv[1:7]]
dev = torch.device("cuda:0")
lib = L.load()
a = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if obf else torch.float32)
for i in range(6):
    L.check(lib.dimx_op_gemm(L.BF16, L.BF16 if obf else L.F32, L.ptr(a), K, L.ptr(w), K, L.ptr(out), N, M, N, K, None,
                             act, None, 0, 0, None, cfg << 8, L.stream_ptr(dev)), "gemm")
torch.cuda.synchronize()
