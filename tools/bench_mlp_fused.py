"""The fused feed-forward sublayer (csrc/mlp_fused.hip) at the headline row count, for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats -d DIR -- python tools/bench_mlp_fused.py [M=76800] [iters=6] [K_o=0|384|768]
(the op packs its weights per call and synchronises: read the KERNEL duration from the trace, not the wall time).
181.2 GFLOP per launch at M = 76 800."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

M = int(sys.argv[1]) if len(sys.argv) > 1 else 76800
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
C, F = 384, 1536
lib = L.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(M, C, device=dev)
w1, b1, w2 = torch.randn(F, C) * C ** -0.5, torch.randn(F) * 0.05, torch.randn(C, F) * F ** -0.5
b2, g, be = torch.randn(C, device=dev) * 0.02, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
hp = lambda t: ctypes.c_void_p(t.data_ptr())
for act, beta in ((2, be), (3, None)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(iters):
        xx = x.clone()
        L.check(lib.dimx_op_mlp_fused(L.ptr(xx), hp(w1), hp(b1), hp(w2), L.ptr(b2), L.ptr(g), L.ptr(beta), M, C, F, act, L.stream_ptr(dev)), "mlp")
    print("act %d ok, finite %s" % (act, bool(torch.isfinite(xx).all())))
