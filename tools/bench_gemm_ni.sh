#!/bin/bash
# GEMM-mode time of the vgg_64 Winograd layers with DVG_GEMM_NI transform positions per workgroup run back to back in one
# software pipeline (one prologue per workgroup instead of one per position)
out=gpurun_out/r04_gemm_ablate
mkdir -p $out
for ni in 0 2 3 4 6 9 12; do
  if [ "$ni" != "0" ]; then export DVG_GEMM_NI=$ni; else unset DVG_GEMM_NI; fi
  echo "=== DVG_GEMM_NI=${ni}" | tee -a $out/ni.txt
  BENCH_BATCHES=${BENCH_BATCHES:-64,576} timeout -k 10 200 python3 tools/bench_wino_parts.py 2>&1 | grep -v amdgpu.ids | sed -E 's/\| out .*//' | grep "gemm\|---" | tee -a $out/ni.txt
done
