"""Fold the MFMA counter pass over one forward (tools/r05_profiles.sh step 2: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, ... over
`bench.py --steps 1`) into profiles/pmc_prefill_mfma_<hash>.json, where dimx.roofline reads `pmc_mfma_busy_pct` of the prefill's two
MFMA kernels (mlp_fused_kernel, attn_tr_kernel).  <hash> = sha256[:12] over csrc/mlp_fused.hip + csrc/attention_tr.hip.
    python tools/pmc_prefill_record.py <pmc_summary text> <commit>"""
import json
import os
import re
import sys

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import roofline

txt, commit = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None
rec = {"commit": commit, "kernel_source_sha256_12": roofline.kernel_source_hash(roofline.PREFILL_MFMA_SOURCES), "kernels": {},
       "note": "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (summed over the "
               "8 XCDs); averages over the launches of one forward at B = 256, T = 300; source: profiles/%s" % os.path.basename(txt)}
cur = None
for ln in open(txt):
    m = re.search(r"(mlp_fused_kernel<[^>]*>|attn_tr_kernel<[^>]*>)", ln)
    if m and not ln.startswith("    "):
        cur = m.group(1)
        rec["kernels"][cur] = {}
        continue
    if not ln.startswith("    "):
        cur = None if not ln.startswith("--") else cur
        continue
    m = re.match(r"\s+(\w+)\s+n=\s*(\d+)\s+avg=([0-9.eE+-]+)", ln)
    if cur and m:
        rec["kernels"][cur][m.group(1)] = float(m.group(3))
        rec["kernels"][cur]["launches"] = int(m.group(2))
for k, v in rec["kernels"].items():
    if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        v["mfma_busy_pct"] = 100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024)
        if "SQ_INSTS_MFMA" in v:
            v["non_mfma_instructions_per_mfma"] = (v.get("SQ_INSTS_VALU", 0.0) + v.get("SQ_INSTS_SALU", 0.0)) / v["SQ_INSTS_MFMA"]
out = os.path.join(roofline.PROFILES, "pmc_prefill_mfma_%s.json" % rec["kernel_source_sha256_12"])
with open(out, "w") as fh:
    json.dump(rec, fh, indent=1)
print(out, {k: round(v.get("mfma_busy_pct", float("nan")), 1) for k, v in rec["kernels"].items()})
