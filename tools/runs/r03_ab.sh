for rep in 1 2; do
for l in libdimx_hip.so libdimx_hip_prev.so; do
echo "== batch 16 $l"
DIMX_LIB=$PWD/dyadic-interaction-modeling_amd/$l timeout 600 python bench.py --batch 16 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-parity-mode --no-train-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
