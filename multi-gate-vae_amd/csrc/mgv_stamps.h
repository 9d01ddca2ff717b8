// In-kernel phase stamps for the diagnostic builds (diag_*.hip, tools/stamps.py): a wave accumulates the
// s_memtime cycles it spends between STAMP points; the product build compiles them away.
#pragma once
#ifdef MGV_STAMPS
// in-kernel phase stamps (diagnostic build only; never quote its run time, read the SHARES)
#define STAMP_DECL unsigned long long st_t0_ = 0, st_acc_[16] = {0}; int st_k_ = 0;
#define STAMP_BEGIN do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0_) :: "memory"); __builtin_amdgcn_sched_barrier(0); st_k_ = 0; } while (0)
#define STAMP(k) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); st_acc_[k] += t_ - st_t0_; st_t0_ = t_; } while (0)
#define STAMP_FLUSH(a) do { if ((threadIdx.x & 63) == 0 && (a).stamps) for (int k_ = 0; k_ < 16; ++k_) atomicAdd((a).stamps + (threadIdx.x >> 6) * 16 + k_, st_acc_[k_]); } while (0)
#else
#define STAMP_DECL
#define STAMP_BEGIN
#define STAMP(k)
#define STAMP_FLUSH(a)
#endif
