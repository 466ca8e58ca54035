#!/bin/bash
# Same-box A/B of the split-K combine: one launch (in-kernel, default) against the two-launch form (DVG_SPLITK_ONE_LAUNCH=0), twice each.
out=gpurun_out/r04_splitk_ab.txt
for v in 1 0 1 0; do
  DVG_SPLITK_ONE_LAUNCH=$v timeout -k 10 300 python3 bench.py --no-train-leg --no-cpu-baseline --no-f32mfma-leg --no-make-gifs-leg --no-extra-legs --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); f = d['families']['dcgan']
print('one_launch=$v  vgg', d['value'], d['ms_per_step'], d['single_chain']['ms_per_step'], ' dcgan', f['value'], f['ms_per_step'], f['single_chain']['ms_per_step'])" | tee -a $out || exit 1
done
