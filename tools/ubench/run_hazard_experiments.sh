#!/bin/bash
# (1) does an EXEC change behind issued MFMAs disturb them; (2) the select form of the FIRST pair's store
# phase (tools/ab_variants.sh conv_igemm2.hip DVG_FIRST_SELECTS 1 2 3) against the shipped library, REPS launches each.
set -e
out=gpurun_out/r04_hazard
mkdir -p $out
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/ubench/mfma_exec_hazard.hip -o $out/mfma_exec_hazard 2>/dev/null
timeout -k 10 120 $out/mfma_exec_hazard > $out/mfma_exec_hazard.txt 2>&1
tail -4 $out/mfma_exec_hazard.txt
for lib in "" tools/_ab/lib_DVG_FIRST_SELECTS_1.so tools/_ab/lib_DVG_FIRST_SELECTS_2.so tools/_ab/lib_DVG_FIRST_SELECTS_3.so; do
  if [ -n "$lib" ]; then export DVG_HIP_LIB=$PWD/$lib; else unset DVG_HIP_LIB; fi
  timeout -k 10 300 python3 tools/ubench/first_pair_selects_repro.py 2>&1 | grep "^lib=" | tee -a $out/first_pair_selects.txt
done
