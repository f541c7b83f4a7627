"""Does the row stride of A / W matter?  (L2 channel camping hypothesis)
Times the decode-step and prefill GEMM shapes with lda = ldw = K + pad elements (bf16), pad in {0, 32, 64, 128}.

    python tools/bench_ldpad.py [decode|prefill|all]
"""
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")


def run(M, N, K, pad, slabs, out_bf16, act, iters):
    ld = K + pad
    a = torch.zeros(M, ld, device=dev, dtype=torch.bfloat16)
    a[:, :K] = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ws = []
    for _ in range(4):
        w = torch.zeros(N, ld, device=dev, dtype=torch.bfloat16)
        w[:, :K] = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        ws.append(w)
    bias = torch.randn(N, device=dev)
    out = torch.empty((8, M, N) if slabs else (M, N), device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    flags = 5 if slabs else 0

    def one(i):
        L.check(lib.dimx_op_gemm(L.BF16, L.BF16 if out_bf16 else L.F32, L.ptr(a), ld, L.ptr(ws[i % 4]), ld, L.ptr(out), N, M,
                                 N, K, L.ptr(bias) if act else None, act, None, 0, 0, None, flags, L.stream_ptr(dev)), "gemm")
    for i in range(8):
        one(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        one(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


DECODE = [("qkv", 256, 2304, 1152, True, False, 0), ("out", 256, 1152, 768, True, False, 0),
          ("q", 256, 768, 1152, True, False, 0), ("ff1", 256, 4608, 1152, False, True, 3),
          ("ff2", 256, 1152, 4608, True, False, 0), ("logits", 256, 512, 1152, True, False, 0)]
PREFILL = [("cross_kv", 76800, 1536, 1152, False, True, 0), ("vq_l1", 76800, 1536, 384, False, True, 2),
           ("vq_l2", 76800, 384, 1536, False, False, 0), ("enc_qkv", 76800, 2304, 384, False, True, 0),
           ("dec_ff1_tf", 76544, 4608, 1152, False, True, 3)]

which = sys.argv[1] if len(sys.argv) > 1 else "all"
cases = (DECODE if which in ("decode", "all") else []) + (PREFILL if which in ("prefill", "all") else [])
for name, M, N, K, slabs, obf, act in cases:
    row = []
    for pad in (0, 32, 64, 128):
        us = run(M, N, K, pad, slabs, obf, act, 200 if M <= 1024 else 20)
        row.append("pad %3d: %8.2f us" % (pad, us))
    tf = 2.0 * M * N * K / 1e6
    print("%-10s M=%-6d N=%-5d K=%-5d  %s   (%.0f MFLOP)" % (name, M, N, K, "  ".join(row), tf), flush=True)
