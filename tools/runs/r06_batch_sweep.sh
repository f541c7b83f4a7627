#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s %8.1f clips/s %8.2f ms per batch  (%.0f us per decode step incl. prefill share)' % (sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step']*1e3/299))" "$1"; }
{
for b in 1 4 16 32 64 128 192 256; do
  timeout 300 python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline --no-best-of-n 2>/dev/null | pr "B=$b T=300 bf16"
done
for b in 1 16 128; do
  timeout 300 python bench.py --mode f32 --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline --no-best-of-n 2>/dev/null | pr "B=$b T=300 f32"
done
} | tee $O/batch_sweep.txt
