cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_x3
rm -rf $out; mkdir -p $out
B="--steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-families --inflight 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_vgg -o vgg -- python3 bench.py --model vgg $B > $out/bench_vgg.log 2>&1 < /dev/null
tail -c 1500 $out/bench_vgg.log
python3 tools/trace_by_grid.py $out/stats_vgg/*kernel_trace.csv > $out/by_grid.txt 2>&1
find $out -name "*kernel_trace.csv" -delete
