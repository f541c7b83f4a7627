cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-train-step > $O/r03_bench_line_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/r03_bench_kernel_stats.csv
rm -rf $O/kt
