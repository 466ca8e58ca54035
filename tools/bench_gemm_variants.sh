#!/bin/bash
# Same-box A/B of the Winograd GEMM-mode tile variants: library (tools/_ab/*.so or the in-tree one) x DVG_GEMM_TW (8: BM=64, 16: BM=128).
# usage: tools/bench_gemm_variants.sh "lib:tw" ...   (lib "default" = in-tree)
out=gpurun_out/r04_gemm_variants
mkdir -p $out
for spec in "$@"; do
  l=${spec%%:*}; tw=${spec##*:}
  if [ "$l" != default ]; then export DVG_HIP_LIB=$PWD/tools/_ab/lib_$l.so; else unset DVG_HIP_LIB; fi
  if [ "$tw" != "-" ]; then export DVG_GEMM_TW=$tw; else unset DVG_GEMM_TW; fi
  echo "=== lib=$l TW=$tw" | tee -a $out/parts.txt
  BENCH_BATCHES=64,576 timeout -k 10 200 python3 tools/bench_wino_parts.py 2>&1 | grep -v amdgpu.ids | sed -E 's/\| out .*//' | grep "gemm\|---" | tee -a $out/parts.txt || exit 1
  timeout -k 10 300 python3 bench.py --no-families --no-train-leg --no-cpu-baseline --no-f32mfma-leg --no-make-gifs-leg --no-extra-legs --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('    rollout', d['value'], d['ms_per_step'])" | tee -a $out/parts.txt || exit 1
done
