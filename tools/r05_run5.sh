#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/e5; mkdir -p $O
for v in 1 0; do
  if [ $v = 1 ]; then export DIMX_NO_WS72=1; else unset DIMX_NO_WS72; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$v -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline > $O/line$v.json 2>/dev/null
  echo "== DIMX_NO_WS72=$v" >> $O/kstat.txt
  python tools/kstat.py $O/kt$v | sort -k5 -n -r | head -14 >> $O/kstat.txt
  rm -rf $O/kt$v
done
cat $O/kstat.txt
