set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/fp && mkdir -p /tmp/fp
timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python bench.py --mode f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/f32prof.log 2>&1
f=$(find /tmp/fp -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r03_f32_mode_kernel_stats.csv && head -30 "$f" < /dev/null
tail -2 gpurun_out/f32prof.log | cut -c1-400
