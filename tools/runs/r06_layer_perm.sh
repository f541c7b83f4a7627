#!/bin/bash
# round 6 experiment: the layer kernel with every clip's 12 heads spread over 12 CUs of its XCD (DIMX_LAYER_PERM=1) -- does a static
# spread shorten the group barriers that wait for the slowest of an XCD's 32 streams?
O=gpurun_out/r06
mkdir -p $O
DIMX_LAYER_PERM=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_s2s.py -x -q -k "layer or chain or generate or graph" 2>&1 | tail -3
B="timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %.1f clips/s %.2f ms  chain_faults %s' % (sys.argv[1], d['value'], d['ms_per_step'], d.get('chain_faults')))" "$1"; }
{
for rep in 1 2 3; do
  $B 2>/dev/null | pr "default"
  DIMX_LAYER_PERM=1 $B 2>/dev/null | pr "DIMX_LAYER_PERM=1"
done
for p in 0 1; do
  echo "== DIMX_LAYER_PERM=$p, stamps of the last decode step (layer 0)"
  DIMX_LAYER_PERM=$p DIMX_LAYER_PROF=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity-mode --no-train-step 2>&1 | grep -A13 "layer-kernel stamps, layer 0"
done
} | tee $O/layer_perm.txt
