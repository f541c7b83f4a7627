#!/bin/bash
# round 6, last call: PMC records + bench lines of the final kernel sources (tools/r06_profiles.sh), the whole GPU suite, smoke, and the
# driver's own command line
bash tools/r06_profiles.sh be9cf8f > gpurun_out/r06_profiles.log 2>&1; echo "profiles rc $?"; tail -3 gpurun_out/r06_profiles.log | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gpu_tests_final.txt 2>&1; tail -4 gpurun_out/r06/gpu_tests_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py ) > gpurun_out/r06/driver_style_bench.json 2> gpurun_out/r06/driver_style_bench.err; tail -4 gpurun_out/r06/driver_style_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r06/driver_style_bench.json').read().strip().splitlines()[-1]); print('driver-style line:', d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['parity_mode']['value'], d['roofline']['frac'], d['cpu_baseline']['value'])"
