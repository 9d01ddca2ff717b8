// Diagnostic build of the bf16x3 level-sweep kernels with in-kernel phase stamps (s_memtime).
// Separate entry points; never used by the product path.  Read the SHARES, not the run time.
#define MGV_STAMPS 1
#define mgv_func_sweep_fwd_x3 mgv_diag_func_sweep_fwd_x3_impl
#define mgv_func_sweep_bwd_x3 mgv_diag_func_sweep_bwd_x3_impl
#define mgv mgv_diag_lvl
#include "func_level_x3.hip"
#undef mgv
