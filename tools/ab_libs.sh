#!/bin/bash
# Same-box A/B of library variants built with `make -C dvg_amd/csrc variant NAME=x DEFS=...` (libdvg_hip_x.so): the vgg_64 rollout
# (three in flight and one chain), each variant twice, interleaved.  usage: tools/ab_libs.sh base x y ...   ("base" = libdvg_hip.so)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
B="--no-families --no-train-leg --no-cpu-baseline --no-f32mfma-leg --no-make-gifs-leg --no-extra-legs --no-roofline --no-check --sustained-s 0 --allow-variant"
for rep in 1 2; do
  for v in "$@"; do
    lib=dvg_amd/csrc/libdvg_hip_$v.so; [ "$v" = base ] && lib=dvg_amd/csrc/libdvg_hip.so
    DVG_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py $B ${AB_ARGS} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$v rep $rep: in flight', d['value'], 'frames/s', d['ms_per_step'], 'ms; single chain', d['single_chain']['ms_per_step'], 'ms')" || exit 1
  done
done
