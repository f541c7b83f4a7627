set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_module.py -q -x -k "hip_training_step" 2>&1 | grep -v "^    \|^$" | tail -25 | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vq.py tests/test_gpu_s2s.py -q -x 2>&1 | tail -4
O=gpurun_out/diag; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-roofline > $O/line.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/kt
head -30 $O/kernel_stats.csv | cut -c1-200
tail -c 300 $O/line.json
