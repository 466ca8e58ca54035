#!/bin/bash
# first GPU pass of round 2: tests, bench (plain + under torchrun with a forced 1-rank RCCL group)
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r2a/bench1.json 2> gpurun_out/r2a/bench1.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r2a/bench1.err
DVG_FORCE_ALLREDUCE=1 DVG_BENCH_FORCE_PG=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-families > gpurun_out/r2a/bench_pg.json 2> gpurun_out/r2a/bench_pg.err; echo "bench_pg rc=$?"
tail -c 600 gpurun_out/r2a/bench_pg.err
