B="python bench.py --no-cpu-baseline --no-parity-mode --no-roofline --steps 4 --warmup 2"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s: %.1f clips/s %.1f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
DIMX_NO_CHAIN=1 $B 2>/dev/null | pr "G1 nochain"
DIMX_NO_CHAIN=1 DIMX_GEN_GROUPS=2 $B 2>/dev/null | pr "G2 nomask"
DIMX_NO_CHAIN=1 DIMX_GEN_GROUPS=2 DIMX_GEN_CUMASK=1 $B 2>/dev/null | pr "G2 cumask"
DIMX_NO_CHAIN=1 DIMX_GEN_GROUPS=2 DIMX_GEN_CUMASK=1 DIMX_SPLIT_TARGET=144 $B 2>/dev/null | pr "G2 cumask split144"
DIMX_NO_CHAIN=1 DIMX_GEN_GROUPS=4 DIMX_GEN_CUMASK=1 DIMX_SPLIT_TARGET=72 $B 2>/dev/null | pr "G4 cumask split72"
$B 2>/dev/null | pr "G1 chain"
