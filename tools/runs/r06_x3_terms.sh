#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
DIMX_NO_X3=1 timeout 300 python tools/r06_x3_terms.py gen exact_f32_mfma $O/t_exact.npz
DIMX_NO_X3=1 DIMX_F32_NO_SPLIT=1 timeout 300 python tools/r06_x3_terms.py gen exact_f32_mfma_other_summation_order $O/t_exact2.npz
timeout 300 python tools/r06_x3_terms.py gen split_bf16_six_products $O/t_x3.npz
DIMX_X3_ABL=4 timeout 300 python tools/r06_x3_terms.py gen split_bf16_three_products $O/t_x33.npz
python tools/r06_x3_terms.py cmp $O/t_exact.npz $O/t_exact2.npz $O/t_x3.npz $O/t_x33.npz | tee $O/x3_terms.json
rm -f $O/t_*.npz
