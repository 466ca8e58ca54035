cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 300 python3 tools/bench_train.py --model dcgan --iters 3 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -o vgg -- python3 tools/bench_train.py --model vgg --iters 2 > gpurun_out/train_vgg.log 2>&1 < /dev/null
grep -v "rocprofv3\|amdgpu.ids" gpurun_out/train_vgg.log | tail -3
head -16 gpurun_out/prof_train/vgg_kernel_stats.csv | cut -c1-160
