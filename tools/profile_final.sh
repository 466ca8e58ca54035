#!/bin/bash
# End-of-round refresh of the evidence that depends on the training kernels (run through gpurun from the repo root; the rollout
# kernel stats / PMC passes are tools/profile_round.sh): the driver's bench line, rocprofv3 kernel stats of one training iteration
# per family, graph-replay timings of the C2 / C4 / C5 shapes, the GP kernels by step group.  Outputs: gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final
rm -rf $out; mkdir -p $out
t0=$SECONDS
timeout 500 python3 bench.py > $out/bench.json 2> $out/bench.err < /dev/null
echo "bench wall $((SECONDS - t0)) s" > $out/bench_wall.txt
for m in vgg dcgan; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_train_$m -o train -- python3 tools/bench_train.py --model $m --iters 2 > $out/train_${m}_under_rocprof.log 2>&1 < /dev/null
done
for cfg in "--model vgg" "--model dcgan" "--model vgg --channels 3 --batch 16 --n_past 2 --n_future 10" "--model dcgan --channels 3 --batch 16 --n_past 2 --n_future 10" "--model vgg --channels 3 --image_width 128 --batch 4 --n_past 4 --n_future 12" "--model dcgan --channels 3 --image_width 128 --batch 4 --n_past 4 --n_future 12"; do
  timeout 400 python3 tools/bench_train.py $cfg --iters 5 --graph 2>> $out/train.err < /dev/null | grep ms_per_iter >> $out/train_graph.jsonl
done
timeout 300 python3 tools/bench_gp.py > $out/gp_step_groups.txt 2>&1
find $out -name "*kernel_trace.csv" -delete
cat $out/bench_wall.txt; tail -c 400 $out/bench.json; cut -c1-120 $out/train_graph.jsonl
