#!/bin/bash
# Board power and clocks while the headline rollouts run (diagnostic): rocm-smi sampled every 0.5 s beside `bench.py --sustained-s 20`.
# usage (on the GPU box): tools/power_trace.sh [vgg|dcgan]  -> gpurun_out/power_<model>.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
m=${1:-vgg}
out=gpurun_out/power_$m.txt
mkdir -p gpurun_out
: > $out
( for i in $(seq 1 90); do
    echo "t=$i $(/opt/rocm/bin/rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | tr -d '\n' | cut -c1-1200)" >> $out
    sleep 0.5
  done ) &
sampler=$!
timeout -k 10 200 python3 bench.py --model $m --no-families --no-train-leg --no-cpu-baseline --no-f32mfma-leg --no-make-gifs-leg \
    --no-extra-legs --no-roofline --no-check --sustained-s 20 > gpurun_out/power_bench_$m.json 2> gpurun_out/power_bench_$m.err
kill $sampler 2>/dev/null
wait $sampler 2>/dev/null
python3 - "$out" <<'PY'
import json, re, sys
rows = []
for ln in open(sys.argv[1]):
    m = re.match(r"t=(\d+) (.*)", ln)
    if not m or not m.group(2).startswith("{"):
        continue
    try:
        d = json.loads(m.group(2))
    except Exception:
        continue
    c = d.get("card0", {})
    pw = next((v for k, v in c.items() if "ower" in k and "W" in k), None)
    sclk = next((v for k, v in c.items() if k.startswith("sclk")), None)
    use = next((v for k, v in c.items() if "GPU use" in k), None)
    rows.append((int(m.group(1)), pw, sclk, use))
for r in rows:
    print("t=%3d  power %s W  sclk %s  use %s %%" % r)
PY
