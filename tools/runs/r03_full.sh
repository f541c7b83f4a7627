set -x
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err; tail -c 1500 gpurun_out/r03_bench_line.json
bash tools/scale_check.sh 1 2>&1 | tail -6
