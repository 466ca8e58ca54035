#!/bin/bash
# Collect the round's rocprof evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench (vgg_64 + dcgan_64 in one process) and of one training iteration, then separate
#   PMC passes (no trace domains mixed in).  Outputs under gpurun_out/prof_round/; tools/pmc_summary.py turns them into
#   profiles/* (see tools/collect_profiles.py).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_round
# a gpurun call is limited to 20 minutes: PART=1 (kernel-trace statistics) and PART=2 (PMC passes, bench line, training benches) in two
# calls; gpurun merges both into gpurun_out/prof_round/.  Default: both.
PART=${PART:-12}
mkdir -p $out
if [[ $PART == *1* ]]; then
# per-kernel durations are taken with ONE rollout in flight (--inflight 1): with several rollouts overlapping, a kernel's
# duration includes the time it shares the chip with other chains' kernels; the default (3 in flight) is profiled beside it.
# --tile-policy energy: the one chain runs the kernels the headline's chains run (bench.py's roofline leg times the same ones)
B="--steps 10 --warmup 3 --no-check --sustained-s 0 --no-cpu-baseline --no-train-leg --no-families --no-f32mfma-leg --no-make-gifs-leg --no-extra-legs --inflight 1 --tile-policy energy"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_vgg -o vgg -- python3 bench.py --model vgg $B > $out/bench_vgg_under_rocprof.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_dcgan -o dcgan -- python3 bench.py --model dcgan $B > $out/bench_dcgan_under_rocprof.log 2>&1 < /dev/null
B3="--steps 12 --warmup 3 --no-check --sustained-s 0 --no-cpu-baseline --no-train-leg --no-families --no-f32mfma-leg --no-make-gifs-leg --no-extra-legs --no-roofline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_vgg_inflight3 -o vgg -- python3 bench.py --model vgg $B3 > $out/bench_vgg_inflight3_under_rocprof.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_dcgan_inflight3 -o dcgan -- python3 bench.py --model dcgan $B3 > $out/bench_dcgan_inflight3_under_rocprof.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_train -o train -- python3 tools/bench_train.py --model vgg --iters 2 > $out/train_vgg_under_rocprof.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_train_dcgan -o train -- python3 tools/bench_train.py --model dcgan --iters 2 > $out/train_dcgan_under_rocprof.log 2>&1 < /dev/null
fi
if [[ $PART == *2* ]]; then
P="--steps 2 --warmup 1 --no-check --sustained-s 0 --no-cpu-baseline --no-graph --no-train-leg --no-families --no-f32mfma-leg --no-make-gifs-leg --no-extra-legs"
for m in vgg dcgan; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${m}_$tag -o pmc -- python3 bench.py --model $m $P > $out/pmc_${m}_$tag.log 2>&1 < /dev/null
  done
done
timeout 400 python3 bench.py > $out/bench.json 2> $out/bench.err < /dev/null
for cfg in "--model vgg" "--model dcgan" "--model vgg --channels 3 --batch 16 --n_past 2 --n_future 10" "--model dcgan --channels 3 --batch 16 --n_past 2 --n_future 10" "--model vgg --channels 3 --image_width 128 --batch 4 --n_past 4 --n_future 12" "--model dcgan --channels 3 --image_width 128 --batch 4 --n_past 4 --n_future 12"; do
  timeout 400 python3 tools/bench_train.py $cfg --iters 5 --graph 2>> $out/train.err < /dev/null | grep ms_per_iter >> $out/train_graph.jsonl
done
fi
find $out -name "*kernel_trace.csv" -delete   # large; the stats / counter files are what gets kept
ls $out | head -40
tail -c 300 $out/bench.json
