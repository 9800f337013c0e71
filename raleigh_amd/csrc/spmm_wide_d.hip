// Interleaved windowed SpMM kernels for dtype RLH_D (one translation unit per element type so the
// instantiations compile in parallel); the code is spmm_wide.inc.
#define RLH_WIDE_DT RLH_D
#define RLH_WIDE_FN wide_spmm_d
#include "spmm_wide.inc"
