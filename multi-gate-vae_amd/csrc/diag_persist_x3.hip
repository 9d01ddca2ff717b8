// Diagnostic build of the persistent sweep kernels with in-kernel phase stamps (s_memtime).
// Separate entry points; never used by the product path.  Read the SHARES, not the run time.
#define MGV_STAMPS 1
#define mgv_func_sweep_fwd_persist_x3 mgv_diag_func_sweep_fwd_persist_x3
#define mgv_func_sweep_bwd_persist_x3 mgv_diag_func_sweep_bwd_persist_x3
#define mgv_sweep_persist_sync_bytes mgv_diag_sweep_persist_sync_bytes
#define mgv_sweep_persist_max_grid mgv_diag_sweep_persist_max_grid
#define mgv_sweep_persist_slab_floats mgv_diag_sweep_persist_slab_floats
#define mgv_sweep_persist_status mgv_diag_sweep_persist_status
#define mgv mgv_diag_persist
#include "sweep_persist_x3.hip"
#undef mgv
