#!/bin/bash
# Same-box A/B of the tile / schedule switches with three rollouts in flight (tools/bench_concurrent.py): the choices were
# tuned for ONE serial chain of launches, where a launch's tail is idle time; with other chains filling it they may differ.
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python tools/bench_concurrent.py --model vgg --steps 60 --inflight 1,3 2>&1 | grep inflight; }
run X=0
run DVG_GEMM_TW=16
run DVG_GEMM_NI=2
run DVG_WINOGRAD_CHAIN=0
run DVG_WINOGRAD=0
run DVG_UPCONV_AS_CONVT=0
run X=0
