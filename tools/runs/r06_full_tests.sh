#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1
tail -15 $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
