#!/bin/bash
# round 6: (1) is a CU mask honoured at all on this platform?  (2) two 128-clip engines on two streams WITHOUT masks: today's kernels
# vs CU-exclusive kernels (DIMX_GEN_EXCL=1: every decode kernel of an engine asks for more than half a CU's LDS), with a kernel-trace
# timeline of what runs beside what
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=tools/ubench/xcdmask_probe
: > $O/xcdmask_probe2.txt
for c in "bits=0-8 64" "bits=0-128 256" "lo256" "all"; do
  timeout 30 $P $c >> $O/xcdmask_probe2.txt 2>&1
  echo "case '$c' exit $?" >> $O/xcdmask_probe2.txt
done
cat $O/xcdmask_probe2.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %.1f clips/s %.2f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
{
$B 2>/dev/null | pr "default"
DIMX_NO_CHAIN=1 $B 2>/dev/null | pr "NO_CHAIN"
DIMX_NO_CHAIN=1 DIMX_GEN_EXCL=1 $B 2>/dev/null | pr "NO_CHAIN EXCL (one engine)"
DIMX_GEN_GROUPS=2 $B 2>/dev/null | pr "GROUPS=2"
DIMX_GEN_GROUPS=2 DIMX_GEN_EXCL=1 $B 2>/dev/null | pr "GROUPS=2 EXCL"
DIMX_GEN_GROUPS=2 DIMX_GEN_EXCL=1 DIMX_NO_WS72=1 $B 2>/dev/null | pr "GROUPS=2 EXCL NO_WS72"
$B 2>/dev/null | pr "default again"
} | tee $O/two_engines_ab.txt
T="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline"
DIMX_GEN_GROUPS=2 rocprofv3 --kernel-trace --output-format csv -d $O/kt_g2 -- $T > /dev/null 2>&1
DIMX_GEN_GROUPS=2 DIMX_GEN_EXCL=1 rocprofv3 --kernel-trace --output-format csv -d $O/kt_g2x -- $T > /dev/null 2>&1
DIMX_NO_CHAIN=1 rocprofv3 --kernel-trace --output-format csv -d $O/kt_g1 -- $T > /dev/null 2>&1
for d in kt_g1 kt_g2 kt_g2x; do echo "==== $d"; python tools/timeline.py $O/$d 0.6; done | tee $O/two_engines_timeline.txt
# the traces themselves are large: keep only the head of one for the column layout
head -3 $(ls $O/kt_g2/*/*kernel_trace.csv | head -1) > $O/kernel_trace_head.txt
rm -rf $O/kt_g1 $O/kt_g2 $O/kt_g2x
