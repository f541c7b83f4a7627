cd $GRAFT_REPO_ROOT
F="--steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode --no-train-step --no-shard-check --no-roofline"
for cfg in "" "DIMX_NO_CHAIN=1" "DIMX_GEN_GROUPS=2" "DIMX_GEN_GROUPS=2 DIMX_GEN_CUMASK=1" "DIMX_GEN_GROUPS=4"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('chain_faults'))"
done
