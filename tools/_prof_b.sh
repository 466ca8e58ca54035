cd "$GRAFT_REPO_ROOT"
out=gpurun_out/wino_b
rm -rf $out; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "winograd or chain" > $out/tests.txt 2>&1 || exit 1
for v in 0 1 2 9; do DVG_WINO_CHAIN_VARIANT=$v BENCH_BATCHES=64,576 python tools/bench_wino_parts.py 2>/dev/null | sed "s/^/V$v /" >> $out/parts.txt; done
DVG_GEMM_TW=16 BENCH_BATCHES=64,576 python tools/bench_wino_parts.py 2>/dev/null | sed "s/^/TW16 /" >> $out/parts.txt
