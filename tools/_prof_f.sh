cd "$GRAFT_REPO_ROOT"
out=gpurun_out/conc; rm -rf $out; mkdir -p $out
for v in 4 3; do DVG_HIP_LIB=$PWD/tools/_ab/lib_DVG_GEMM_WGS_PER_CU_$v.so python tools/bench_concurrent.py --model vgg --inflight 1,2,3,4 2>/dev/null | sed "s/^/WGS$v /" >> $out/conc.txt; done
