// Diagnostic build of the register-resident bf16x3 backward with in-kernel phase stamps (s_memtime).
// Separate entry points; never used by the product path.  Read the SHARES, not the run time.
#define MGV_STAMPS 1
#define mgv_struct_stage_bwd2_x3 mgv_diag_struct_stage_bwd2_x3_impl
#define mgv_struct_stage_bwd2_ws_floats mgv_diag_struct_stage_bwd2_ws_floats
#define mgv mgv_diag_b2
#include "struct_stage_bwd2_x3.hip"
#undef mgv
