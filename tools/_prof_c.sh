cd "$GRAFT_REPO_ROOT"
out=gpurun_out/wino_c
rm -rf $out; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $out/tests.txt 2>&1 || { tail -20 $out/tests.txt; exit 1; }
BENCH_BATCHES=64,576 python tools/bench_wino_parts.py > $out/parts.txt 2>/dev/null
python tools/bench_layers.py > $out/layers.txt 2>/dev/null
python bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err
