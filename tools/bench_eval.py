"""Wall time of the reference's test-time protocol end to end (x_engine_pt.evaluate_test_epoch: best of 10 generations per clip by
Frechet distance) with the Frechet distances on the host (the reference's scipy arithmetic) and on the device.
    python tools/bench_eval.py [B=256] [T=300] [batches=3]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L
from dimx import prng, x_engine_pt
from dimx.seq2seq_pretrain import SLMFT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
model = SLMFT(numeric_mode=L.MODE_PERF_BF16).to(dev).eval()
batches = []
for i in range(NB):
    src = torch.from_numpy(prng.normal(40 + i, "ev.src", (B, T, 824)))
    tgt = torch.from_numpy(prng.normal(40 + i, "ev.tgt", (B, T, 56)))
    batches.append((src, tgt, [T] * B, None, ["c%d_%d" % (i, j) for j in range(B)]))
x_engine_pt.evaluate_test_epoch(model, batches[:1], dev, beam_size=10, seed=5, fd_backend="device")     # warm-up
res = {}
for backend in ("reference", "device"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    yt, yp, xs, ids = x_engine_pt.evaluate_test_epoch(model, batches, dev, beam_size=10, seed=5, fd_backend=backend)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[backend] = yp
    print("evaluate_test_epoch fd_backend=%-9s %d batches of %d clips x 10 tries: %6.2f s  (%.1f clips/s end to end)" % (
        backend, NB, B, dt, NB * B / dt), flush=True)
same = sum(int(np.array_equal(a, b)) for a, b in zip(res["reference"], res["device"]))
print("same winner for %d of %d clips" % (same, len(res["reference"])))
