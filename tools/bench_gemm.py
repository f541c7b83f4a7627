"""GEMM microbenchmark on the GPU box: every dense shape of the C3 workload x tile/stage configs.
Host launch overhead (~8 us per ctypes call) hides kernels shorter than that, so run it under rocprofv3 and
read the GPU-side durations:

    rocprofv3 --kernel-trace --output-format csv -d OUT -o g -- python tools/bench_gemm.py decode PLAN.json
    python tools/bench_gemm.py --parse OUT PLAN.json
"""
import csv
import glob
import json
import math
import sys


def parse(outdir, plan_path):
    plan = json.load(open(plan_path))
    rows = []
    for f in glob.glob(outdir + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    i = 0
    cur = None
    line = []
    for label, cfgname, warm, n in plan:
        i += warm
        d = [rows[j][1] for j in range(i, i + n)]
        i += n
        if label != cur:
            if line:
                print("%-22s" % cur + "  ".join(line))
            cur, line = label, []
        d.sort()
        line.append("%s:%.1f" % (cfgname, d[len(d) // 2] / 1e3))
    if line:
        print("%-22s" % cur + "  ".join(line))
    assert i == len(rows), (i, len(rows))


if len(sys.argv) > 1 and sys.argv[1] == "--parse":
    parse(sys.argv[2], sys.argv[3])
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, ".")
import dimx  # noqa
from dimx import lib as L  # noqa: E402

dev = torch.device("cuda:0")
lib = L.load()
plan = []


def run(label, M, N, K, out_bf16, inplace, cfg, splitk, act=0, iters=12, ncopies=6, accumulate=False):
    a = (torch.randn(M, K, device=dev)).bfloat16()
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(ncopies)]
    res = torch.randn(M, N, device=dev) if inplace else None
    out = res if inplace else torch.zeros(M, N, device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    flags = (1 if (inplace or accumulate) else 0) | (cfg << 8) | (splitk << 16) | ((4 if accumulate else 0))
    warm = 3
    for i in range(warm + iters):
        L.check(lib.dimx_op_gemm(L.BF16, L.BF16 if out_bf16 else L.F32, L.ptr(a), K, L.ptr(ws[i % ncopies]), K,
                                 L.ptr(out), N, M, N, K, None, act, L.ptr(res), N if inplace else 0, 0, None, flags,
                                 L.stream_ptr(dev)), "gemm")
    torch.cuda.synchronize()
    plan.append((label, "c%d/s%d" % (cfg, splitk), warm, iters))


which = sys.argv[1] if len(sys.argv) > 1 else "all"
plan_path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/plan.json"
if which in ("decode", "all"):
    shapes = [("qkv N2304 K1152", 2304, 1152, False, False, 0, True), ("self_out N1152 K768", 1152, 768, False, True, 0, False),
              ("cross_q N768 K1152", 768, 1152, False, False, 0, True), ("ff1 N4608 K1152", 4608, 1152, True, False, 3, False),
              ("ff2 N1152 K4608", 1152, 4608, False, True, 0, False), ("logits N512 K1152", 512, 1152, False, False, 0, True)]
    for name, N, K, obf, inpl, act, accum in shapes:
        for cfg in (4, 14):
            for sp in ((1, 2, 4, 8) if (inpl or accum) else (0,)):
                run(name, 256, N, K, obf, inpl, cfg, sp, act, accumulate=accum)
if which in ("prefill", "all"):
    shapes = [("vq_qk N768 K384", 768, 384, True, False, 0), ("vq_out N384 K384", 384, 384, False, True, 0),
              ("vq_l1 N1536 K384", 1536, 384, True, False, 2), ("vq_l2 N384 K1536", 384, 1536, False, True, 0),
              ("xe_qkv N2304 K384", 2304, 384, True, False, 0), ("xe_out N384 K768", 384, 768, False, True, 0),
              ("ckv N1536 K1152", 1536, 1152, True, False, 0)]
    for name, N, K, obf, inpl, act in shapes:
        for cfg in (14, 18, 19):
            run(name, 76800, N, K, obf, inpl, cfg, 0, act, iters=6, ncopies=2)
if which == "kscan":
    # main-loop rate vs fixed (prologue + epilogue) cost: same M, N at two K; plus a square reference shape
    for name, M, N, K, obf in (("ckv K1152 bf16", 76800, 1536, 1152, True), ("ckv K4608 bf16", 76800, 1536, 4608, True),
                               ("ckv K1152 f32", 76800, 1536, 1152, False), ("sq 8192 bf16", 8192, 8192, 8192, True),
                               ("sq 4096 bf16", 4096, 4096, 4096, True)):
        for cfg in (1, 14, 18, 19):
            run(name, M, N, K, obf, False, cfg, 0, 0, iters=5, ncopies=2)
json.dump(plan, open(plan_path, "w"))
