"""Do an HBM-bound kernel (decode attention) and latency-bound small GEMMs overlap when issued on two streams?"""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import engine as E
from dimx import lib as L

dev = torch.device("cuda:0")
lib = L.load()
B, H, T = 256, 12, 300
kc = [torch.randn(B, H, 304, 64, device=dev).bfloat16() for _ in range(4)]
vc = [torch.randn(B, H, 304, 64, device=dev).bfloat16() for _ in range(4)]
q = torch.randn(B, H * 64, device=dev).bfloat16()
a = torch.randn(128, 1152, device=dev).bfloat16()
ws = [(torch.randn(2304, 1152, device=dev) / 34).bfloat16() for _ in range(4)]
out = torch.empty(8, 128, 2304, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def attn(n, stream):
    with torch.cuda.stream(stream):
        for i in range(n):
            E.op_decode_attn(q, kc[i % 4], vc[i % 4], T, 0.125)


def gemms(n, stream):
    with torch.cuda.stream(stream):
        for i in range(n):
            L.check(lib.dimx_op_gemm(L.BF16, L.F32, L.ptr(a), 1152, L.ptr(ws[i % 4]), 1152, L.ptr(out), 2304, 128, 2304,
                                     1152, None, 0, None, 0, 0, None, 5, L.stream_ptr(dev)), "g")


def timed(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


for _ in range(2):
    attn(20, s1); gemms(20, s2)
N = 200
ta = timed(lambda: attn(N, s1))
tg = timed(lambda: gemms(4 * N, s2))
tb = timed(lambda: (attn(N, s1), gemms(4 * N, s2)))
print("attention alone %.2f ms, gemms alone %.2f ms, both on two streams %.2f ms (sum %.2f, max %.2f)" % (ta, tg, tb, ta + tg, max(ta, tg)))
