#!/bin/bash
# round 6, VERDICT item 5: does the layer kernel pay for fetching its three 1.77 MB weight slices once per XCD (8 x, 38 MB per launch)?
# The same launch with the slice fetches removed (DIMX_LAYER_ABL=1: wrong results, same streams / barriers / MFMAs), interleaved.
for rep in 1 2 3; do
  for abl in 0 1; do
    DIMX_LAYER_ABL=$abl python - <<PY
import os, sys, torch
sys.path.insert(0, ".")
import dimx
from dimx import roofline as R
r = R.layer_chain(256, 300, torch.device("cuda:0"), iters=200)
print("weight slices %-12s %6.2f us per launch  (%.0f GB/s of the algorithmic bytes)" % ("NOT fetched" if os.environ.get("DIMX_LAYER_ABL") == "1" else "fetched", r["avg_launch_us"], r["achieved"]))
PY
  done
done
