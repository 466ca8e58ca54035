cd "$GRAFT_REPO_ROOT"
out=gpurun_out/abl2; rm -rf $out; mkdir -p $out
for v in 0 4 5; do DVG_HIP_LIB=$PWD/tools/_ab/lib_DVG_ABLATE_$v.so BENCH_BATCHES=64,576 python tools/bench_wino_parts.py 2>/dev/null | grep -E "c4.1|c3.1|c2.1" | awk -v v=$v '{printf "A%s %s %s %s", v,$1,$2,$3; for(i=4;i<=NF;i++) if ($i=="gemm") printf " | gemm %s us %s TF", $(i+1),$(i+3); printf "\n"}' >> $out/gemm.txt; done
