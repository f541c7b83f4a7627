#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
python -m pytest tests/test_gpu_gemm_x3.py -x -q 2>&1 | tail -15
python -m pytest tests/test_gpu_s2s.py tests/test_gpu_configs.py tests/test_gpu_module.py -x -q 2>&1 | tail -15
B="python bench.py --mode f32 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-train-step"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %.1f clips/s %.2f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
{
DIMX_NO_X3=1 $B 2>/dev/null | pr "f32 mode, f32 MFMA decode GEMMs (DIMX_NO_X3=1)"
$B 2>/dev/null | pr "f32 mode, split-bf16 decode GEMMs"
} | tee $O/x3_ab.txt
