"""One prefill GEMM shape, a few launches, for rocprofv3 --pmc passes (MFMA / LDS / wait counters).
    python tools/gemm_pmc.py M N K cfg [iters]"""
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")
M, N, K, cfg = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 6
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for i in range(iters):
    L.check(lib.dimx_op_gemm(L.BF16, L.BF16, L.ptr(a), K, L.ptr(w), K, L.ptr(out), N, M, N, K, None, 0, None, 0, 0, None,
                             cfg << 8, L.stream_ptr(dev)), "gemm")
torch.cuda.synchronize()
print("done")
