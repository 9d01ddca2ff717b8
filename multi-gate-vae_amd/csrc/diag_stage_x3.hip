// Diagnostic build of the bf16x3 struct-stage kernels with in-kernel phase stamps (s_memtime).
// Separate entry points; never used by the product path.  Read the SHARES, not the run time.
#define MGV_STAMPS 1
#define mgv_struct_stage_fwd_x3 mgv_diag_struct_stage_fwd_x3_impl
#define mgv_struct_stage_bwd_x3 mgv_diag_struct_stage_bwd_x3_impl
#define mgv mgv_diag
#include "struct_stage_x3.hip"
#undef mgv
