cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -2
for tag in new; do
O=gpurun_out/r03t_$tag; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline > $O/line.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/stats.csv
rm -rf $O/kt
done
for e in "DIMX_QKV_VT=1" "X=1" "DIMX_QKV_VT=1" "X=1"; do
echo "== $e"
env $e timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-train-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
