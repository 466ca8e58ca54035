mkdir -p gpurun_out/r06
export DVG_DP_SHARE_GPU=1 DVG_DP_BACKEND=gloo
port=29801
run2() { port=$((port+1)); timeout -k 10 280 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port tools/diag_repeat_backward.py --model vgg --batch 4 --repeats 4 "$@" 2>&1 | grep "rank 0. checksum\|repeat [0-9]*:\|Error\|error" | sort | cut -c1-200; }
echo "#### pack entries under contention"; timeout -k 10 250 python3 tools/diag_pack_repeat.py --noise train --explain 1 2>&1 | grep -v amdgpu.ids | tail -5
echo "#### one process + another trainer, caches dropped every repeat"; timeout -k 10 250 python3 tools/diag_repeat_backward.py --model vgg --batch 4 --repeats 40 --drop_caches --noise proctrain 2>&1 | grep "diag_repeat\|repeat [0-9]*:" | tail -4 | cut -c1-250
echo "#### two ranks, --meet gloo, 20 launches"; for i in $(seq 20); do run2 --meet gloo; done | sort | uniq -c
echo "#### two ranks, --meet gloo --drop_caches"; for i in 1 2 3; do run2 --meet gloo --drop_caches; done | sort | uniq -c
