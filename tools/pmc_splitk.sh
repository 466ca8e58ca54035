#!/bin/bash
# HBM-side traffic of dcgan_64's split-K layers with the one-launch combine (DVG_SPLITK_ONE_LAUNCH=1) against the finish launch:
# FETCH_SIZE / WRITE_SIZE passes of the same command as tools/profile_round.sh (separate --pmc runs, kernel trace only).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_splitk
rm -rf $out; mkdir -p $out
P="--steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-train-leg --no-families --no-f32mfma-leg --no-make-gifs-leg --no-extra-legs"
export DVG_SPLITK_ONE_LAUNCH=1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_dcgan_$c -o pmc -- python3 bench.py --model dcgan $P > $out/pmc_$c.log 2>&1 < /dev/null || exit 1
done
python3 tools/pmc_summary.py $out > $out/summary.json
find $out -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_splitk/summary.json"))
for k, v in d.items():
    if k.startswith("conv_igemm2<1") or k.startswith("splitk_finish"):
        n = v["FETCH_SIZE"]["dispatches"]
        print(k, n, "traffic MB/launch", round((2 * v["FETCH_SIZE"]["avg"] + v["WRITE_SIZE"]["avg"]) * 1024 / 1e6, 1))
PY
