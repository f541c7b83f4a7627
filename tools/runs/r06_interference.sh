#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
python tools/r06_interference.py > $O/interference.txt 2>&1
cat $O/interference.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %.1f clips/s %.2f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
{
DIMX_GEN_GROUPS=2 DIMX_GEN_CUMASK=1 $B 2>/dev/null | pr "GROUPS=2 CUMASK (16 CUs of every XCD)"
DIMX_GEN_GROUPS=2 DIMX_GEN_CUMASK=1 DIMX_GEN_EXCL=1 $B 2>/dev/null | pr "GROUPS=2 CUMASK EXCL"
} | tee -a $O/two_engines_ab.txt
