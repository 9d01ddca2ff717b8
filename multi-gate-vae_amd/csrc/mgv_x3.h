// Split-precision ("bf16x3") matrix products: an fp32 operand x is carried as two bf16 planes
// hi = bf16(x), lo = bf16(x - hi) and a product as  lo*hi + hi*lo + hi*hi  on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  The dropped lo*lo term and the 2^-17 truncation
// of the split leave ~1e-5 relative error per product (measured on the reference fixtures: losses
// move by ~1e-6, embeddings by ~3e-5) at 3/16 of the fp32-MFMA cost.
#pragma once
#include "mgv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace mgv {

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

__device__ __forceinline__ void split4(const float4& v, bf16x4& hi, bf16x4& lo) {
    __bf16 h0, h1, h2, h3, l0, l1, l2, l3;
    split_bf16(v.x, h0, l0); split_bf16(v.y, h1, l1); split_bf16(v.z, h2, l2); split_bf16(v.w, h3, l3);
    hi = bf16x4{h0, h1, h2, h3};
    lo = bf16x4{l0, l1, l2, l3};
}

__device__ __forceinline__ bf16x8 ldfrag(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
// the same from a pointer known to be GLOBAL memory although the compiler cannot see it (a pointer that went through an opaque
// asm statement is generic: its loads become flat_load, which also occupy the LDS counter and queue)
__device__ __forceinline__ bf16x8 ldfrag_global(const __bf16* p) {
    return *(const __attribute__((address_space(1))) bf16x8*)p;
}
__device__ __forceinline__ void st_bf4(__bf16* p, const bf16x4& v) { *reinterpret_cast<bf16x4*>(p) = v; }

__device__ __forceinline__ f32x4 mfma_bf16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// c += (a_hi + a_lo) . (b_hi + b_lo) over a 32-deep k-step, small terms first
__device__ __forceinline__ void mma_x3(f32x4& c, const bf16x8& ahi, const bf16x8& alo, const bf16x8& bhi, const bf16x8& blo) {
    c = mfma_bf16(alo, bhi, c);
    c = mfma_bf16(ahi, blo, c);
    c = mfma_bf16(ahi, bhi, c);
}

// Transposed MFMA operand fragment straight from a ROW-major bf16 plane: the 8 k-values a lane needs are 8
// consecutive ROWS (k0 + 8q .. +7) of ONE column (c0 + r).  ds_read_b64_tr_b16 hands lane i of a 16-lane
// group column i of a 4-row x 16-column block (lane 4q'+p supplies the address of block row q', columns
// 4p..4p+3; checked on hardware by tools/test_tr_read.hip), so two of them make the fragment and no
// transposed copy of the tile is ever written.  All 64 lanes must be active (the ISA requires full EXEC).
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 ldfrag_tr(const __bf16* plane, int ld, int k0, int c0) {
    const int lane = threadIdx.x & 63, q = lane >> 4, i16 = lane & 15;
    const __bf16* p = plane + (k0 + 8 * q + (i16 >> 2)) * ld + c0 + 4 * (i16 & 3);
    const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4v*)p);
    const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4v*)(p + 4 * ld));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// The same for MFMAs whose BOTH operands are read this way (weight gradients: dG^T x X^T, k = the tile's rows).  The k-slots of a
// lane are then free to be ANY rows as long as both operands agree, and the rows are dealt so that each half of the instruction is
// bank-conflict free: the LDS serves a 64-lane b64 read in two passes of 32 lanes (256 B); with the K + 8 plane rows (4 x odd dwords)
// the eight rows {8q .. 8q+3 : q = 0, 1} of ldfrag_tr's first pass overlap on 24 banks (every pass takes two cycles:
// SQ_LDS_BANK_CONFLICT was 42 % of the struct backward's LDS-active cycles, profiles/r04_pmc_mfma.json), while eight rows of equal
// parity tile the 64 banks exactly.  Lane group q reads rows b + {0, 2, 4, 6} and b + 16 + {0, 2, 4, 6}, b = 8 (q & 1) + (q >> 1):
// pass one (q = 0, 1) the even rows 0..14, pass two (q = 2, 3) the odd ones; every row of the 32-deep k-step exactly once.
__device__ __forceinline__ bf16x8 ldfrag_tr2(const __bf16* plane, int ld, int k0, int c0) {
    const int lane = threadIdx.x & 63, q = lane >> 4, i16 = lane & 15;
    const __bf16* p = plane + (k0 + 8 * (q & 1) + (q >> 1) + 2 * (i16 >> 2)) * ld + c0 + 4 * (i16 & 3);
    const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4v*)p);
    const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4v*)(p + 16 * ld));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// bf16 LDS planes: [rows][K] with K+8 elements per row (144-byte rows at K=64: the 16 rows a
// ds_read_b128 fragment touches start 36 dwords apart, i.e. on distinct 4-bank groups)
template <int K>
struct Plane {
    static constexpr int LD = K + 8;
    static constexpr int ELEMS = kTileRows * LD;
};

}  // namespace mgv
