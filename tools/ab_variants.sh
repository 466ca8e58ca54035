#!/bin/bash
# Build variants of libdvg_hip.so that differ in one -D macro of one source file, for same-box A/B runs:
#   tools/ab_variants.sh conv_igemm2.hip DVG_VMEM_POLICY 0 1 2     ->  tools/_ab/lib_<macro>_<value>.so
# Select one at run time with DVG_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../dvg_amd/csrc"
make -s -j6
src=$1; macro=$2; shift 2
mkdir -p ../../tools/_ab
# The WRONG-results knobs (DVG_ABLATE, DVG_X3_TERMS, DVG_FIRST_SELECTS) need EXTRA=-DDVG_TIMING_EXPERIMENTS=1 (dvg_common.h
# refuses them otherwise); dvg_build_info() of a variant carries the macro, so bench.py can tell it from the product.
srcid=$(make -s srcid)
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D${macro}=${v} $EXTRA -c $src -o /tmp/ab_${macro}_${v}.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D${macro}=${v} $EXTRA -DDVG_SRC_ID="\"$srcid\"" \
      -DDVG_VARIANT_NAME="\"${macro}=${v}\"" -c build_info.hip -o /tmp/ab_${macro}_${v}_info.o
  objs=$(ls *.o | grep -v "^${src%.hip}.o$" | grep -v "^build_info.o$")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/ab_${macro}_${v}.o /tmp/ab_${macro}_${v}_info.o -o ../../tools/_ab/lib_${macro}_${v}.so
  echo built tools/_ab/lib_${macro}_${v}.so
done
