cd "$GRAFT_REPO_ROOT"
for cfg in "--model dcgan --channels 3 --batch 16 --n_past 2 --n_future 10" "--model vgg --channels 3 --batch 16 --n_past 2 --n_future 10" "--model dcgan --channels 3 --image_width 128 --batch 4 --n_past 4 --n_future 12" "--model vgg --channels 3 --image_width 128 --batch 4 --n_past 4 --n_future 12"; do
  timeout 400 python3 tools/bench_train.py $cfg --iters 10 --graph 2>> gpurun_out/train_shapes.err < /dev/null | grep ms_per_iter | cut -c1-200 >> gpurun_out/train_shapes.jsonl
done
cat gpurun_out/train_shapes.jsonl
