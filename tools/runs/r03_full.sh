set -x
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4
timeout 600 python __graft_entry__.py smoke | tail -2
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
tail -c 400 gpurun_out/r03_bench_line.json
