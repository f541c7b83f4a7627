cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/tp && mkdir -p /tmp/tp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -- python tools/bench_train.py 16 300 3 f32 > gpurun_out/trainprof.log 2>&1
f=$(find /tmp/tp -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r03_train_step_f32_kernel_stats.csv
