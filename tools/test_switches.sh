#!/bin/bash
# The GPU test suite under every environment switch of README.md (run through gpurun from the repo root).
# A gpurun call is limited to 20 minutes: pass a subset of switches as arguments to split the matrix over several calls.
# The tests marked `slow` (the oracle at B = 50 on the host cores: minutes of CPU work that no switch changes) run in the default
# suite only; DVG_HIP_LIB=.../libdvg_hip_f32mfma.so (the native f32-MFMA build of the library) gets the FULL suite.
# pytest writes straight into a file under gpurun_out/ (gpurun kills a command that stays silent for 7 minutes).
mkdir -p gpurun_out/switches
sum=gpurun_out/switches/summary_$(date +%s).txt   # one per call: gpurun merges by file name
if [ $# -gt 0 ]; then set -- "$@"; else set -- "DVG_HIP_LIB=$PWD/dvg_amd/csrc/libdvg_hip_f32mfma.so" "DVG_WINOGRAD=0" "DVG_WINOGRAD=2" "DVG_WINOGRAD_CHAIN=0" \
         "DVG_WINOGRAD_CHAIN=1" "DVG_WINOGRAD_CHAIN=2" "DVG_TIME_BATCH=0" "DVG_TIME_BATCH=1" \
         "DVG_WINOGRAD_WGRAD=0" "DVG_SKIP_HOIST=0" "DVG_UPCONV_AS_CONVT=0"; fi
for v in "$@"; do
  echo "== $v" | tee -a $sum
  sel="gpu and not slow"; case "$v" in DVG_HIP_LIB=*) sel="gpu";; esac
  log=gpurun_out/switches/$(echo "$v" | tr -c 'A-Za-z0-9=_\n' '_').txt
  env $v timeout 1100 python -m pytest tests -q -m "$sel" -x > "$log" 2>&1
  tail -4 "$log" | grep -v "^$\|Docs:\|warnings.html" | tee -a $sum
done
