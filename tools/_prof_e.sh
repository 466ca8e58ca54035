cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_e
rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_vgg -o vgg -- python3 bench.py --model vgg --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-families --inflight 1 --no-roofline > $out/bench_vgg.log 2>&1 < /dev/null
f=$(find $out/trace_vgg -name "*kernel_trace.csv" | head -1)
python3 tools/trace_by_grid.py $f 13 > $out/vgg_by_grid.txt 2>&1
python3 tools/trace_gaps.py $f > $out/vgg_gaps.txt 2>&1
find $out -name "*kernel_trace.csv" -delete
