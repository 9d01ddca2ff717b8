// Grid-wide barrier for persistent kernels on MI355X (8 XCDs, one L2 each, a vector L1 per CU that other CUs' stores never
// refresh).  XCD-hierarchical, after MI355X_MICROARCH.md "barrier-xcd": every workgroup adds to its XCC's counter; the last arrival
// of an XCC adds to the top counter, waits for all XCCs there and then publishes the XCC's generation word, which the other
// workgroups of that XCC poll.  Visibility of the data exchanged across the barrier (cdna_hip_programming.md, Guideline 16):
//   * producers store their payload WRITE-THROUGH (sc1: st4_wt / st1_wt below), every storing wave drains `s_waitcnt vmcnt(0)`,
//     then the workgroup's barrier, then ONE lane arrives (template flag RELEASE adds the XCC leader's agent-scope release fence,
//     for kernels that publish with plain stores);
//   * consumers: ONE relaxed poll per workgroup, ONE agent-scope acquire (buffer_inv sc1: drops this CU's L1), its vmcnt(0), the
//     workgroup's barrier, then plain vector loads.
// Nothing depends on dispatch order or on which XCD a workgroup landed: the XCC id only groups arrivals, a census at kernel start
// counts the workgroups per XCC.  Every spin is bounded by wall-clock time (s_memrealtime, 100 MHz); a spin that gives up writes its
// code to `error`, every other spin sees that word and leaves too, and the kernel ends.  State words are monotonic within a launch
// and zeroed by the launcher (hipMemsetAsync) before every launch.  The grid MUST be co-resident (the launchers size it by CUs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgv {

constexpr int kGbLine = 32;                       // 32 words: every counter on a 128-byte line of its own
constexpr int kGbXcc = 8;
constexpr unsigned long long kGbTimeoutTicks = 300000000ull;      // 3 s of the 100 MHz constant clock

struct GridBarState {
    unsigned xcc_cnt[kGbXcc][kGbLine];
    unsigned xcc_gen[kGbXcc][kGbLine];
    unsigned census[kGbXcc][kGbLine];
    unsigned top_cnt[kGbLine];
    unsigned all_cnt[kGbLine];
    unsigned error[kGbLine];                      // != 0: a spin gave up (its code); read back by the host (mgv_sweep_persist_status)
};
static_assert(sizeof(GridBarState) % 16 == 0, "the memset block is a multiple of 16 bytes");

struct GridBarLocal { unsigned x, n_x, n_xcc, k; };

__device__ __forceinline__ unsigned gb_load(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned gb_add(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gb_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ unsigned gb_xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x & (kGbXcc - 1);
}

// one lane polls *p until it reaches `target` (wrap-safe); false: gave up (timeout, or another workgroup's error word)
__device__ __forceinline__ bool gb_wait_ge(unsigned* p, unsigned target, GridBarState* st, unsigned code) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 1;; ++spins) {
        if ((int)(gb_load(p) - target) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 127u) == 0u) {
            if (gb_load(&st->error[0]) != 0u) return false;
            if (wall_clock64() - t0 > kGbTimeoutTicks) { gb_store(&st->error[0], code); return false; }
        }
    }
}

// Kernel start, every thread calls it: census of the workgroups per XCC behind one flat arrival counter.  `flag` is one LDS word.
__device__ __forceinline__ bool grid_barrier_init(GridBarState* st, unsigned grid, GridBarLocal& L, int* flag) {
    L.x = 0; L.n_x = 1; L.n_xcc = 1; L.k = 0;
    if (threadIdx.x == 0) {
        L.x = gb_xcc_id();
        unsigned old = gb_add(&st->census[L.x][0], 1u);
        asm volatile("" : "+v"(old));              // the census add has returned before the arrival below is issued
        gb_add(&st->all_cnt[0], 1u + (old & 0u));
        const bool ok = gb_wait_ge(&st->all_cnt[0], grid, st, 1u);
        unsigned n_xcc = 0;
        for (int i = 0; i < kGbXcc; ++i) {
            const unsigned c = gb_load(&st->census[i][0]);
            n_xcc += c != 0u;
            if (i == (int)L.x) L.n_x = c;
        }
        L.n_xcc = n_xcc;
        *flag = ok ? 1 : 0;
    }
    __syncthreads();
    return *flag != 0;
}

// Every thread of every workgroup calls it the same number of times.  false: a spin gave up — leave the kernel.
template <bool RELEASE>
__device__ __forceinline__ bool grid_barrier(GridBarState* st, GridBarLocal& L, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // EVERY storing wave: its payload stores have left this CU
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok;
        L.k += 1u;
        const unsigned old = gb_add(&st->xcc_cnt[L.x][0], 1u);
        if (old + 1u == L.k * L.n_x) {                         // last arrival of this XCC: every workgroup of it has drained
            if (RELEASE) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // buffer_wbl2 sc1: this XCD's dirty L2 lines
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (asm: the compiler may drop the fence's own wait)
            }
            gb_add(&st->top_cnt[0], 1u);
            ok = gb_wait_ge(&st->top_cnt[0], L.k * L.n_xcc, st, 2u);
            if (ok) gb_store(&st->xcc_gen[L.x][0], L.k);
        } else {
            ok = gb_wait_ge(&st->xcc_gen[L.x][0], L.k, st, 3u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // buffer_inv sc1: this CU's L1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *flag = ok ? 1 : 0;
    }
    __syncthreads();
    return *flag != 0;
}

// write-through (sc1) stores of handed-off payload: the bytes leave the XCD's L2 at once, no release fence is needed
typedef unsigned gb_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gb_rsrc(const void* base, uint64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ void st4_wt(__amdgpu_buffer_rsrc_t rs, uint64_t byte_off, const float4& v) {
    const gb_u32x4 u = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)(unsigned)byte_off, 0, 16);
}
__device__ __forceinline__ void st1_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace mgv
