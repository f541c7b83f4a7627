# prefill as clip groups on several streams (default: automatic = 4 at C3) against one batch on one stream (DIMX_PREFILL_GROUPS=1), interleaved
cd $GRAFT_REPO_ROOT
F="--steps 20 --warmup 3 --no-cpu-baseline --no-parity-mode --no-train-step --no-shard-check --no-roofline"
for rep in 1 2 3; do
  for cfg in "DIMX_PREFILL_GROUPS=1" "DIMX_PREFILL_GROUPS=2" ""; do
    v=$(env $cfg timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.2f faults %d' % (d['value'], d['ms_per_step'], d['chain_faults']))")
    echo "rep $rep ${cfg:-automatic (4 groups)  } $v"
  done
done
