"""Fold the MFMA / TCC counter passes over the fused cross-attention K/V projection (tools/attic/r03_profiles.sh step 4) into
profiles/pmc_gemm256_<kernel-source-hash>.json, the file dimx.roofline.cross_kv_gemm reads `pmc_mfma_busy_pct` from.
    python tools/pmc_gemm256_record.py <pmc_summary text> <line file of the same run> <commit>"""
import json
import os
import re
import sys

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import roofline

txt, line_file, commit = sys.argv[1], sys.argv[2], sys.argv[3]
vals, launches, on = {}, 0, False
for ln in open(txt):
    if "gemm256p2_kernel" in ln or "gemm256_kernel" in ln:
        on = True
        continue
    m = re.match(r"\s+(\w+)\s+n=\s*(\d+)\s+avg=([0-9.eE+-]+)", ln)
    if on and m:
        vals[m.group(1)] = float(m.group(3))
        launches = int(m.group(2))
    elif on and not ln.startswith(" "):
        on = False
us = None
m = re.search(r"'avg_launch_us': ([0-9.]+)", open(line_file).read())
if m:
    us = float(m.group(1))
cyc = vals["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
rec = {"kernel": "gemm256p2_kernel<dimx::bf16, 0, true> (cross-attention K/V projection, 4 layers per launch, M=76800 N=6144 K=1152)",
       "commit": commit, "kernel_source_sha256_12": roofline.kernel_source_hash("gemm256.hip")}
rec.update(vals)
rec["launches"] = launches
rec["mfma_busy_pct"] = 100.0 * vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
if us:
    rec["shader_clock_GHz_under_pmc"] = cyc / (us * 1e3)
    rec["avg_launch_us_under_pmc"] = us
rec["note"] = ("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs); kernel cycles = GRBM_GUI_ACTIVE / 8; source: "
               "profiles/%s, launch time under the PMC pass from profiles/%s" % (os.path.basename(txt), os.path.basename(line_file)))
out = os.path.join(roofline.PROFILES, "pmc_gemm256_%s.json" % rec["kernel_source_sha256_12"])
with open(out, "w") as fh:
    json.dump(rec, fh, indent=1)
print(out, "mfma_busy_pct %.1f" % rec["mfma_busy_pct"])
