#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if ! timeout 300 python -m pytest tests/test_gpu_gemm_x3.py -x -q > $O/x3_tests.txt 2>&1; then tail -20 $O/x3_tests.txt; echo 'x3 tests FAILED: stopping'; exit 1; fi
tail -2 $O/x3_tests.txt
bash tools/runs/r06_x3_abl.sh > /dev/null 2>&1
grep -A12 "as built" $O/x3_gemm_abl.txt | grep "x3\|=="
B="timeout 300 python bench.py --mode f32 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-train-step --no-mode-compare"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %.1f clips/s %.2f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
{
DIMX_NO_X3=1 $B 2>/dev/null | pr "f32 mode, f32 MFMA decode GEMMs (DIMX_NO_X3=1)"
$B 2>/dev/null | pr "f32 mode, split-bf16 decode GEMMs"
} | tee $O/x3_ab3.txt
timeout 600 python -m pytest tests/test_gpu_s2s.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
