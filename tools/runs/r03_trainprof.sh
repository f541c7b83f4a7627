set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_legacy.py -x -q -s -k "train_epoch" 2>&1 | tail -8
rm -rf /tmp/tp && mkdir -p /tmp/tp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -- python tools/bench_train.py 16 300 3 bf16 > gpurun_out/trainprof.log 2>&1
f=$(find /tmp/tp -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r03_train_step_kernel_stats.csv && head -30 "$f" < /dev/null
