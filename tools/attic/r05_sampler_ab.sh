# kernel-trace averages of the decode step's tail kernels (sampler, final chain) + a headline line, one gpurun call
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/samp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/samp/prof -o s -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-train-step --no-shard-check --no-roofline > /dev/null 2>&1
f=$(find gpurun_out/samp/prof -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-200 > gpurun_out/samp/stats_head.csv
grep -E "sample_kernel|xcd_chain|add_slabs|xcd_layer" $f | cut -c1-60,180-400
rm -rf gpurun_out/samp/prof
timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-mode --no-train-step --no-shard-check --no-roofline > gpurun_out/samp/line.json 2> gpurun_out/samp/err.txt
cut -c1-200 gpurun_out/samp/line.json
