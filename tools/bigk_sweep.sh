#!/bin/bash
# lgemm reduction-slice sweep on the big-K short GEMMs, cold weights (measurement only)
for uc in 0 1; do for f in none 1,0,0,0,0,1 1,0,0,0,0,2 1,0,0,0,0,4 1,0,0,0,0,8 1,0,0,0,0,16; do
  echo "== force $f uc $uc"
  if [ $f = none ]; then ADNM_LG_UC=$uc COLD=1 SHAPES=tools/data/gemm_shapes_bigk.txt NOLIB=1 PRECS=bf16 REPS=20 python tools/kbench_gemm.py 2>&1 | grep -v amdgpu.ids | cut -c1-90
  else ADNM_LG_UC=$uc COLD=1 ADNM_SK_FORCE=$f SHAPES=tools/data/gemm_shapes_bigk.txt NOLIB=1 PRECS=bf16 REPS=20 python tools/kbench_gemm.py 2>&1 | grep -v amdgpu.ids | cut -c1-90; fi
done; done
