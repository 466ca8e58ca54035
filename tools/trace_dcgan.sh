#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_${1:-dcgan}
rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o tr -- python3 bench.py --model ${1:-dcgan} --steps 20 --warmup 3 --no-cpu-baseline --no-families --no-train-leg --no-roofline --no-f32mfma-leg --no-make-gifs-leg --no-extra-legs > $out/bench.log 2>&1 < /dev/null
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $f ${2:-burst} > $out/gaps.txt
head -40 $out/gaps.txt
rm -f $f
