"""Run only the two roofline microbenchmarks of bench.py (for rocprofv3 --pmc passes)."""
import json
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import roofline

dev = torch.device("cuda:0")
print(json.dumps({"layer": roofline.layer_chain(256, 300, dev, iters=20),
                  "attn": roofline.decode_attention(256, 300, "bf16", dev, iters=20),
                  "gemm": roofline.decode_gemm(256, "bf16", dev, iters=20)}))
