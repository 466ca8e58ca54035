#!/bin/bash
# Collect the round's rocprof evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench (vgg_64) and of dcgan_64 and one training iteration, then three separate
#   PMC passes (no trace domains mixed in).  Outputs under gpurun_out/prof_round/; tools/pmc_summary.py + the
#   snippet in DESIGN.md section 5 turn them into profiles/*.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_round
rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o vgg -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_under_rocprof.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_dcgan -o dcgan -- python3 bench.py --model dcgan --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_dcgan_under_rocprof.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_train -o train -- python3 tools/bench_train.py --model vgg --iters 2 > $out/train_vgg_under_rocprof.log 2>&1 < /dev/null
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$tag -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $out/pmc_$tag.log 2>&1 < /dev/null
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmcd_$c -o pmc -- python3 bench.py --model dcgan --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $out/pmcd_$c.log 2>&1 < /dev/null
done
timeout 300 python3 bench.py > $out/bench_vgg.json 2> $out/bench_vgg.err < /dev/null
timeout 200 python3 bench.py --model dcgan > $out/bench_dcgan.json 2> $out/bench_dcgan.err < /dev/null
for cfg in "--model vgg" "--model dcgan" "--model vgg --channels 3 --batch 16 --n_past 2 --n_future 10" "--model dcgan --channels 3 --batch 16 --n_past 2 --n_future 10" "--model vgg --channels 3 --image_width 128 --batch 4 --n_past 4 --n_future 12" "--model dcgan --channels 3 --image_width 128 --batch 4 --n_past 4 --n_future 12"; do
  timeout 400 python3 tools/bench_train.py $cfg --iters 5 --graph 2>> $out/train.err < /dev/null | grep ms_per_iter >> $out/train_graph.jsonl
done
rm -f $out/*/*kernel_trace.csv   # large; the stats / counter files are what gets kept
ls -la $out $out/stats | head -40
tail -c 300 $out/bench_vgg.json
