timeout 900 python tools/bench_eval.py 256 300 3 2>&1 | grep -v amdgpu
