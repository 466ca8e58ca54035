#!/bin/bash
# The GPU test suite under every environment switch of README.md (run through gpurun from the repo root).
# A gpurun call is limited to 20 minutes: pass a subset of switches as arguments to split the matrix over several calls.
mkdir -p gpurun_out/switches
# (DVG_HIP_LIB=.../libdvg_hip_f32mfma.so runs the suite on the native f32-MFMA build of the library.)
if [ $# -gt 0 ]; then set -- "$@"; else set -- "DVG_HIP_LIB=$PWD/dvg_amd/csrc/libdvg_hip_f32mfma.so" "DVG_WINOGRAD=0" "DVG_WINOGRAD=2" "DVG_WINOGRAD_CHAIN=0" "DVG_WINOGRAD_CHAIN=1" \
         "DVG_UPCONV_WINOGRAD=0" "DVG_FIRST_PAIR=0" "DVG_NO_SPLITK=1" "DVG_TIME_BATCH=0" "DVG_TIME_BATCH=1" "DVG_GEMM_NT=2" "DVG_WINOGRAD_WGRAD=0" "DVG_SAVE_WINO_V=0" "DVG_DENSE_BATCH=1" \
         "DVG_DIRECT_GRADS=0" "DVG_WGRAD_BATCH=1" "DVG_SKIP_HOIST=0" "DVG_UPCONV_AS_CONVT=0" "DVG_LATENT_STREAM=0" "DVG_GP_THREADS=256" "DVG_FUSED_ELBO=0" "DVG_LSTM_SEQ=0" "DVG_SPLITK_ONE_LAUNCH=1" "DVG_SHARE_PREFIX=0"; fi
for v in "$@"; do
  echo "== $v"
  env $v timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | grep -v "^$\|Docs:\|warnings.html"
done 2>&1 | tee -a gpurun_out/switches/summary.txt
