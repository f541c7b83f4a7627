"""A/B of gemm256 tuning knobs on the cross-attention K/V projection as dimx_encode_ctx launches it (M = 76800, N = 6144,
K = 1152, bf16): every configuration (a set of DIMX_G256_* environment variables, see csrc/gemm256.hip) in its own process,
interleaved rounds, then one DIMX_G256_PROF pass each (in-kernel interval sums of block 0, waves 0 and 4).
    python tools/g256_var.py [rounds] ["VAR=3 DESYNC=1000,0" ...]
"""
import os
import subprocess
import sys

CHILD = r"""
import sys, torch
sys.path.insert(0, '.')
import dimx
from dimx import roofline
import os
if os.environ.get('DIMX_G256_ZERO'):
    _rn = torch.randn
    torch.randn = lambda *a, **k: torch.zeros(*a, **k)
r = roofline.cross_kv_gemm(256, 300, 'bf16', torch.device('cuda:0'), iters=%d)
print('RES %%.1f %%.2f' %% (r['avg_launch_us'], r['util_pct']))
"""


def run(cfg, iters, extra=None):
    env = dict(os.environ)
    for kv in cfg.split():
        k, v = kv.split("=")
        env["DIMX_G256_" + k] = v
    if extra:
        env.update(extra)
    p = subprocess.run([sys.executable, "-c", CHILD % iters], env=env, capture_output=True, text=True)
    res = [ln for ln in p.stdout.splitlines() if ln.startswith("RES")]
    prof = [ln for ln in p.stderr.splitlines() if ln.startswith("g256prof")]
    if not res:
        return None, p.stderr[-600:]
    us, pct = res[-1].split()[1:]
    return (float(us), float(pct)), prof


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    cfgs = sys.argv[2:] or ["VAR=3", "VAR=4", "VAR=4 ZERO=1"]
    prof_cfgs = [c for c in cfgs if "VAR=4" not in c]
    acc = {c: [] for c in cfgs}
    for r in range(rounds):
        for c in cfgs:
            res, info = run(c, 12)
            if res is None:
                print("[%s] failed: %s" % (c, info), flush=True)
                continue
            acc[c].append(res)
            print("round %d [%s]: %.1f us  %.2f %% of 2.5 PF" % (r, c, res[0], res[1]), flush=True)
    for c in cfgs:
        if acc[c]:
            us = sorted(x[0] for x in acc[c])
            med = us[len(us) // 2]
            print("[%s]: median %.1f us = %.2f %%, best %.1f us" % (c, med, 2.0 * 76800 * 6144 * 1152 / med / 1e6 / 2500 * 100, us[0]))
    if os.environ.get("G256_PROF_TOO", "1") == "1":
        for c in prof_cfgs:
            res, prof = run(c, 2, {"DIMX_G256_PROF": "1"})
            print("prof [%s] (%s):" % (c, res))
            for ln in (prof or [])[-2:]:
                print("   ", ln)
