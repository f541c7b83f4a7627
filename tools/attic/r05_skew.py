"""Is the attention-phase skew of xcd_layer_kernel systematic (the same CUs late every launch) or random?
Per-block 'cross attention done' stamps of several launches: correlation between launches, and by XCD / CU slot."""
import sys
import torch
sys.path.insert(0, ".")
import dimx  # noqa
from dimx import roofline

dev = torch.device("cuda:0")
runs = []
for r in range(4):
    prof = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
    roofline.layer_chain(256, 300, dev, iters=8 + r, prof=prof)
    p = prof.view(256, 16).cpu().double()
    t0 = p[:, 0].min()
    runs.append(((p[:, 9] - p[:, 8]) / 100.0, (p[:, 1] - p[:, 0]) / 100.0))   # cross-attention phase, self-attention phase per block
cross = torch.stack([a for a, _ in runs])
selfa = torch.stack([b for _, b in runs])
for name, x in (("cross", cross), ("self", selfa)):
    print(name, "phase us per block: mean %.2f  min %.2f  max %.2f  std over blocks %.2f" % (x.mean(), x.min(), x.max(), x.std(1).mean()))
    c = torch.corrcoef(x)
    print("  correlation of the per-block times between launches:", [round(float(c[0, i]), 2) for i in range(1, 4)])
    byx = x.mean(0).view(32, 8)   # block b = slot * 8 + xcd-ish (blockIdx & 7)
    print("  mean by blockIdx & 7:", [round(float(v), 2) for v in byx.mean(0)])
    print("  slowest 8 blocks (mean over launches):", sorted(((round(float(v), 2), int(i)) for i, v in enumerate(x.mean(0))), reverse=True)[:8])
