#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/e14; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_gpu_s2s.py -x -q -k "layer_kernel" 2>&1 | tail -3 > $O/tests.txt
cat $O/tests.txt
DIMX_LAYER_PROF=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity-mode --no-train-step 2>&1 | grep -A14 "layer 1" | head -16 > $O/stamps.txt
cat $O/stamps.txt
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export DIMX_NO_LAYER_CHAIN=1; else unset DIMX_NO_LAYER_CHAIN; fi
  echo "DIMX_NO_LAYER_CHAIN=$v" >> $O/bench.txt
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-140 >> $O/bench.txt
done
cat $O/bench.txt
