"""Wall time of the hand-written HIP training step (dimx.train_hip.HipTrainer: forward + backward + clip + AdamW) next to the
PyTorch-autograd restatement it replaced (dimx.train, rocBLAS / hipBLASLt + autograd), same model, same batch.
    python tools/bench_train.py [B=16] [T=300] [steps=5] [which=all|bf16|f32|hip]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L
from dimx import prng
from dimx import train as T
from dimx.seq2seq_pretrain import SLMFT
from dimx.train_hip import HipTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Tn = int(sys.argv[2]) if len(sys.argv) > 2 else 300
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
which = sys.argv[4] if len(sys.argv) > 4 else "all"
dev = torch.device("cuda:0")
v_s = torch.from_numpy(prng.normal(1, "bt.vs", (B, Tn, 56))).to(dev)
v_l = torch.from_numpy(prng.normal(1, "bt.vl", (B, Tn, 56))).to(dev)
v_a = torch.from_numpy(prng.normal(1, "bt.va", (B, Tn, 768))).to(dev)
mask = torch.ones(B, Tn, dtype=torch.bool, device=dev)


def timed(fn):
    for _ in range(3):   # kernel by kernel, capture of the step graph, first replay
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for mode, name in ((L.MODE_PERF_BF16, "bf16"), (L.MODE_PARITY_F32, "f32")):
    if which not in ("all", "hip", name):
        continue
    m = SLMFT(numeric_mode=mode).to(dev)
    with torch.no_grad():
        _, z = m.forward_vq(v_s, v_l, mask, with_speaker=False)
    tr = HipTrainer(m, lr=1e-5, clip=1.0)
    ms = timed(lambda: tr.train_step(v_s, v_l, v_a, mask, kv_mask=False, z_l=z))
    print("HIP training step  %-4s B=%d T=%d: %8.1f ms  (%.1f clips/s)   [graph replays %d, kernel-by-kernel steps %d, graph nodes %d]"
          % ((name, B, Tn, ms, B / ms * 1e3) + tr.graph_stats()), flush=True)
    del tr, m
    torch.cuda.empty_cache()
if which != "all":
    sys.exit(0)
m = SLMFT().to(dev)
m.train()
with torch.no_grad():
    _, z = m.forward_vq(v_s, v_l, mask, with_speaker=False)
opt = T.make_optimizer(m, lr=1e-5)
with torch.enable_grad():
    ms = timed(lambda: T.train_step(m, opt, v_s, v_l, v_a, mask, clip=1.0, kv_mask=False))
print("autograd restatement (f32 torch ops) B=%d T=%d: %8.1f ms  (%.1f clips/s)" % (B, Tn, ms, B / ms * 1e3))
