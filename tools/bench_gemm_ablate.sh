#!/bin/bash
# GEMM-mode timing of the vgg_64 Winograd layers under the ablation builds (tools/ab_variants.sh conv_igemm2.hip DVG_ABLATE 1 2 6 7):
# what the staging of each operand costs the launch.  Timing only - the ablated builds compute wrong products.
out=gpurun_out/r04_gemm_ablate
mkdir -p $out
for v in "" 1 2 6 7; do
  if [ -n "$v" ]; then export DVG_HIP_LIB=$PWD/tools/_ab/lib_DVG_ABLATE_$v.so; else unset DVG_HIP_LIB; fi
  echo "=== DVG_ABLATE=${v:-0}" | tee -a $out/parts.txt
  timeout -k 10 200 python3 tools/bench_wino_parts.py 2>&1 | grep -v amdgpu.ids | sed -E 's/\| out .*//' | tee -a $out/parts.txt
done
