"""Phase breakdown of the XCD-local chain kernels (csrc/chain.hip) from in-kernel wall-clock stamps (100 MHz).
    DIMX_CHAIN_PROF=1 python tools/chain_phases.py
Weights rotate over 8 copies so that a launch never finds its slice in L2 (as in the decode loop, where 129 MB of
weights pass through the 8 x 4 MB L2s every step)."""
import os
import sys

import torch

os.environ["DIMX_CHAIN_PROF"] = "1"
sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L

lib = L.load()
dev = torch.device("cuda:0")
B, C, K1 = 256, 1152, 768
NAMES = ["start", "burst1 landed", "mfma1", "reduce+store1", "barrier1", "row phase", "barrier2", "A2/W2 landed", "mfma2",
         "reduce+store2"]      # the deferred forms (A-ln, B-ln) have no row phase / barrier 2 stamps


def run(kind, iters=24, rotate=8):
    bf = torch.bfloat16
    a1 = torch.randn(B, K1, device=dev).to(bf)
    w1 = [(torch.randn(C, K1, device=dev) / K1 ** 0.5).to(bf) for _ in range(8)]
    n2 = 768 if kind in ("A", "A-ln") else 512
    w2 = [(torch.randn(n2, C, device=dev) / C ** 0.5).to(bf) for _ in range(8)]
    gamma = torch.ones(C, device=dev)
    slabs = torch.randn(4, B, C, device=dev) * 0.1
    y = torch.empty(B, C, device=dev, dtype=bf)
    out2 = torch.empty(B, n2, device=dev)
    x = torch.randn(B, C, device=dev)
    scratch = torch.zeros(512 + B * C + 256 * 16 * 2, dtype=torch.int32, device=dev)
    acc = torch.zeros(256, 16, dtype=torch.float64)
    n = 0
    defer = kind in ("A-ln", "B-ln")       # deferred LayerNorm: what generate() launches for the attention out-projections
    stats = torch.zeros(8, 32, 32, 2, device=dev)
    cs2 = torch.randn(n2, device=dev)
    for i in range(iters):
        g1 = kind in ("A", "B")
        g2 = kind in ("A", "C")
        if defer:
            w2p = L.ptr(w2[i % rotate]) if kind == "A-ln" else None
            L.check(lib.dimx_op_chain_ln(L.ptr(a1), K1, L.ptr(w1[i % rotate]), L.ptr(x), L.ptr(y), L.ptr(stats), w2p,
                                         L.ptr(cs2) if w2p else None, n2 if w2p else 0, L.ptr(out2) if w2p else None, B, C,
                                         L.ptr(scratch), L.stream_ptr(dev)), "op_chain_ln")
        else:
            L.check(lib.dimx_op_chain(L.ptr(a1) if g1 else None, K1 if g1 else 0, L.ptr(w1[i % rotate]) if g1 else None, L.ptr(x),
                                      L.ptr(slabs) if kind == "C" else None, 4 if kind == "C" else 0, L.ptr(gamma), L.ptr(y),
                                      L.ptr(w2[i % rotate]) if g2 else None, n2 if g2 else 0, L.ptr(out2) if g2 else None, B, C,
                                      L.ptr(scratch), L.stream_ptr(dev)), "op_chain")
        torch.cuda.synchronize()
        assert int(scratch[129]) == 0
        st = scratch[512 + B * C:].view(torch.int64).view(256, 16).cpu().double()
        if i >= 8:
            acc += st - st[:, :1].min()      # relative to the earliest block start
            n += 1
    acc /= n
    print("chain kind %s, %d weight copies in rotation (stamps in us relative to the first block's start; mean / max over "
          "the 256 blocks)" % (kind, rotate))
    prev = None
    for j, name in enumerate(NAMES):
        col = acc[:, j] / 100.0
        if col.mean() < -1e6 or (col.abs().sum() == 0 and j > 0):
            continue
        print("  %-16s mean %6.2f  max %6.2f%s" % (name, col.mean(), col.max(),
                                                    "" if prev is None else "   (+%.2f)" % (col.mean() - prev)))
        prev = col.mean()


for k in ("A", "B", "C", "A-ln", "B-ln"):
    run(k)
run("A", rotate=1)      # weights already in the XCD's L2: what an L2 prefetch by the preceding kernel could buy
