for abl in 0 8 1 2 4 3 7; do echo "== abl $abl"; DIMX_G256_ABL=$abl python tools/bench_prefill.py 0 2>&1 | grep -v amdgpu | sed -n '1p;6p'; done
