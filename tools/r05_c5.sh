#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c5; mkdir -p $O
python bench.py --steps 2 --warmup 1 --batch 64 --frames 1500 --no-cpu-baseline --no-parity-mode --no-train-step 2>/dev/null | tail -1 > $O/r05_bench_c5_shard.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --steps 2 --warmup 1 --batch 64 --frames 1500 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline > $O/line_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/r05_c5_kernel_stats.csv
rm -rf $O/kt
head -c 1500 $O/r05_bench_c5_shard.json; echo
head -25 $O/r05_c5_kernel_stats.csv | cut -c1-200
