mkdir -p gpurun_out/r2g
export BENCH_MIN_C=128
for v in "" "DVG_GEMM_NI=1" "DVG_GEMM_NI=2" "DVG_GEMM_NI=4" "DVG_GEMM_TW=16" "DVG_GEMM_TW=16 DVG_GEMM_NI=1" "DVG_GEMM_TW=16 DVG_GEMM_NI=2"; do
  echo "== $v"; env $v timeout 200 python tools/bench_winograd.py 2>&1 | grep -v amdgpu.ids | sed 's/direct.*| F4/F4/'
done > gpurun_out/r2g/ab_ni.txt 2>&1
cat gpurun_out/r2g/ab_ni.txt
