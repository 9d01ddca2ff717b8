// Diagnostic build of the front / back pipelined bf16x3 backward with in-kernel phase stamps (s_memtime).
// Separate entry points; never used by the product path.  Read the SHARES, not the run time.
#define MGV_STAMPS 1
#define mgv_struct_stage_bwd2_x3 mgv_diag3_struct_stage_bwd2_x3_impl
#define mgv_struct_stage_bwd2_ws_floats mgv_diag3_struct_stage_bwd2_ws_floats
#define mgv_diag_set_stamps2 mgv_diag3_set_stamps2
#define mgv_struct_stage_bwd3_x3 mgv_diag_struct_stage_bwd3_x3_impl
#define mgv mgv_diag_b3
#include "struct_stage_bwd2_x3.hip"
#include "struct_stage_bwd3_x3.hip"
#undef mgv
