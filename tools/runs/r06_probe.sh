#!/bin/bash
# round 6, first GPU call: baseline line of the tree as round 5 left it, then the whole-XCD CU-mask probe (one process per case,
# each under its own timeout: a mask that leaves an XCD without CUs might never dispatch that XCD's workgroups)
mkdir -p gpurun_out/r06
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r06/baseline_line.json 2> gpurun_out/r06/baseline_line.err
tail -c 600 gpurun_out/r06/baseline_line.json
P=tools/ubench/xcdmask_probe
for c in all lo hi lo256 graph both "xs=55" "xs=03 64" all; do
  timeout 30 $P $c >> gpurun_out/r06/xcdmask_probe.txt 2>&1
  echo "case '$c' exit $?" >> gpurun_out/r06/xcdmask_probe.txt
done
cat gpurun_out/r06/xcdmask_probe.txt
