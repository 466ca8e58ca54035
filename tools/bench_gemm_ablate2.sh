#!/bin/bash
# The bare loop of the GEMM mode: DVG_ABLATE 2 (no staging), 3 (+ no barriers), 8 (2 + 1/16 of the M store), 9 (3 + 1/16 of the M store)
out=gpurun_out/r04_gemm_ablate
mkdir -p $out
for v in "" 2 3 8 9; do
  if [ -n "$v" ]; then export DVG_HIP_LIB=$PWD/tools/_ab/lib_DVG_ABLATE_$v.so; else unset DVG_HIP_LIB; fi
  echo "=== DVG_ABLATE=${v:-0}" | tee -a $out/parts2.txt
  BENCH_BATCHES=64,576 timeout -k 10 200 python3 tools/bench_wino_parts.py 2>&1 | grep -v amdgpu.ids | sed -E 's/\| out .*//' | grep "gemm\|---" | tee -a $out/parts2.txt
done
