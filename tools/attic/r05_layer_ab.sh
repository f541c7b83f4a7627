# interleaved A/B on one box: the layer kernel against the four launches it replaces (DIMX_NO_LAYER_CHAIN=1), 20 timed steps each
cd $GRAFT_REPO_ROOT
F="--steps 20 --warmup 3 --no-cpu-baseline --no-parity-mode --no-train-step --no-shard-check --no-roofline"
for rep in 1 2 3; do
  for cfg in "" "DIMX_NO_LAYER_CHAIN=1"; do
    v=$(env $cfg timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.2f' % (d['value'], d['ms_per_step']))")
    echo "rep $rep ${cfg:-layer kernel          } $v"
  done
done
