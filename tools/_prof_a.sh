cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_a
rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_vgg -o vgg -- python3 bench.py --model vgg --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-families --inflight 1 --no-roofline > $out/bench_vgg.log 2>&1 < /dev/null
f=$(find $out/trace_vgg -name "*kernel_trace.csv" | head -1)
python3 tools/trace_by_grid.py $f 10 > $out/vgg_by_grid.txt 2>&1
head -3 $f > $out/trace_head.txt
find $out -name "*kernel_trace.csv" -delete
for t in 256 512 1024; do DVG_GP_THREADS=$t python3 tools/bench_gp.py 2>/dev/null | sed "s/^/T=$t /" >> $out/gp_threads.txt; done
