"""Where does a decode GEMM block spend its time?  In-kernel wall-clock stamps (100 MHz) of the 64x64 LDS-DMA kernel
(ABL = 3 instantiation) on the ff1 shape (M=256, N=4608, K=1152, GELU, bf16 out, no split-K).
    DIMX_GEMM_PROF=1 python tools/gemm_phases.py"""
import os
import sys

import torch

os.environ["DIMX_GEMM_PROF"] = "1"
sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")
M, N, K = 256, 4608, 1152
a = torch.randn(M, K, device=dev).bfloat16()
ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(6)]
bias = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
nblk = (M // 64) * (N // 64)
prof = torch.zeros(nblk * 32, dtype=torch.int64, device=dev)
acc = torch.zeros(nblk, 32, dtype=torch.float64)
n = 0
for i in range(20):
    prof.zero_()
    L.check(lib.dimx_op_gemm(L.BF16, L.BF16, L.ptr(a), K, L.ptr(ws[i % 6]), K, L.ptr(out), N, M, N, K, L.ptr(bias), 3,
                             L.ptr(prof), N, 0, None, 3 << 8, L.stream_ptr(dev)), "gemm")
    torch.cuda.synchronize()
    st = prof.view(nblk, 32).cpu().double()
    if i >= 6:
        acc += st - st[:, :1].min()
        n += 1
acc /= n * 100.0
names = ["start", "prologue DMAs issued"] + ["iteration %d begins" % i for i in range(18)]
print("ff1 M=256 N=4608 K=1152, 64x64 tiles, 288 blocks; us relative to the first block's start (mean / max over blocks)")
prev = None
for j in list(range(20)) + [28, 29]:
    col = acc[:, j]
    name = names[j] if j < 20 else ("main loop done" if j == 28 else "epilogue stored")
    print("  %-24s mean %6.2f  max %6.2f%s" % (name, col.mean(), col.max(), "" if prev is None else "  (+%.2f)" % (col.mean() - prev)))
    prev = col.mean()
