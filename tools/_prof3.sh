cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_train_x3
rm -rf $out; mkdir -p $out
python3 tools/bench_train.py --model vgg --iters 4 2>/dev/null | tail -1
python3 tools/bench_train.py --model dcgan --iters 4 2>/dev/null | tail -1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o train -- python3 tools/bench_train.py --model vgg --iters 2 > $out/train.log 2>&1 < /dev/null
find $out -name "*kernel_trace.csv" -delete
head -16 $out/stats/train_kernel_stats.csv | cut -c1-150
