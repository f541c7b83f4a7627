cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_s2s.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_legacy.py tests/test_gpu_slm.py tests/test_gpu_kernels.py -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
python bench.py --mode f32 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
