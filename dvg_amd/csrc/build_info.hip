// dvg_build_info(): which build of the library is loaded (ABI 9).  The Makefile recompiles this file whenever any source of
// the library changes and passes DVG_SRC_ID = the first 12 hex digits of sha256(all sources, in Makefile order), so a loaded
// .so can be matched to the tree it was built from (`make -C dvg_amd/csrc srcid` prints the tree's id; tests/test_abi.py
// compares the two).  The knob values are those THIS translation unit was compiled with - `make variant` hands the same
// DEFS to every translation unit.
#include "dvg_common.h"

#ifndef DVG_SRC_ID
#define DVG_SRC_ID "unknown"
#endif
#ifndef DVG_VARIANT_NAME
#define DVG_VARIANT_NAME ""
#endif
#define DVG_STR2(x) #x
#define DVG_STR(x) DVG_STR2(x)

extern "C" const char* dvg_build_info(void) {
    return "abi=9 bf16x3=" DVG_STR(DVG_BF16X3) " x3_terms=" DVG_STR(DVG_X3_TERMS) " ablate=" DVG_STR(DVG_ABLATE)
           " first_selects=" DVG_STR(DVG_FIRST_SELECTS) " timing_experiments=" DVG_STR(DVG_TIMING_EXPERIMENTS)
           " variant=" DVG_VARIANT_NAME " src=" DVG_SRC_ID;
}
