#!/bin/bash
# Rollouts in flight: frames/s of both families for 1 ... 6 chains (same box, back to back).
out=gpurun_out/r04_inflight.txt
for m in vgg dcgan; do
  for n in 1 2 3 4 5 6; do
    timeout -k 10 200 python3 bench.py --model $m --inflight $n --no-families --no-train-leg --no-cpu-baseline --no-f32mfma-leg --no-make-gifs-leg --no-extra-legs --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$m inflight=$n', d['value'], d['ms_per_step'])" | tee -a $out || exit 1
  done
done
