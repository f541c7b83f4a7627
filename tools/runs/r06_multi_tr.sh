#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_s2s.py -x -q -k "multi_sample" -s 2>&1 | tail -6
B="timeout 300 python bench.py --steps 3 --warmup 1 --samples 10 --no-cpu-baseline --no-parity-mode --no-roofline --no-train-step"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.1f sequences/s %.2f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
{
for rep in 1 2; do
DIMX_NO_MULTI_TR=1 $B 2>/dev/null | pr "VALU multi-query kernel"
$B 2>/dev/null | pr "attention_tr (MFMA)"
done
} | tee $O/multi_tr_ab.txt
rm -rf $O/kts
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kts -- python bench.py --steps 2 --warmup 1 --samples 10 --no-cpu-baseline --no-parity-mode --no-roofline --no-train-step > /dev/null 2>&1
cp $(ls $O/kts/*/*kernel_stats.csv | head -1) $O/samples10_kernel_stats_tr.csv
rm -rf $O/kts
head -8 $O/samples10_kernel_stats_tr.csv | cut -c1-150
