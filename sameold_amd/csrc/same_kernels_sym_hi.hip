// same_kernels_sym_hi.hip -- the 44.1 / 48 kHz instantiations of the symbol-paced pipeline as a translation unit of their own:
// same_kernels_sym.hip once more with SYM_TU_HI, compiled with the scheduler set for instruction-level parallelism
// (sameold_amd/build.py: SOURCE_FLAGS; the why and the measurements are in same_kernels_sym.hip beside SYM_SPLIT_TU).
#define SYM_TU_HI 1
#include "same_kernels_sym.hip"
