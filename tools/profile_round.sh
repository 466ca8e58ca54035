#!/bin/bash
# Collect the round's rocprof evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench, then three separate PMC passes (no trace domains mixed in).
# Outputs under gpurun_out/prof_round/; tools/pmc_summary.py turns them into profiles/*.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_round
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o vgg -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_under_rocprof.log 2>&1 < /dev/null
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$tag -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $out/pmc_$tag.log 2>&1 < /dev/null
done
find $out -name "*.csv" | head -20
tail -1 $out/bench_under_rocprof.log | cut -c1-300
