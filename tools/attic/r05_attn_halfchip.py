"""Round 5: how much of the HBM stream rate do the decode attention's CUs reach when only part of the chip streams?
The cross-attention form at T = 300 for B clips (one block = one clip = 12 waves above 128 clips; two waves per head at or below)."""
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import roofline as R

dev = torch.device("cuda:0")
for B in (32, 64, 96, 128, 136, 160, 192, 224, 256):
    r = R.decode_attention(B, 300, "bf16", dev, iters=300)
    print("B %3d  %6.2f us  %7.1f GB/s  per clip-CU %5.1f GB/s  %s" % (B, r["avg_launch_us"], r["achieved"], r["achieved"] / B, r["kernel"][:60]))
