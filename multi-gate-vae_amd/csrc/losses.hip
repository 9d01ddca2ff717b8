// Edge / pair reductions of the DG_AE train step: inner-product decoder, reconstruction loss with
// confusion counters, functional-similarity loss (cosine distance -> z-normalisation -> L1),
// reparameterisation sampler + KL.  All HBM/gather bound: H/4 lanes read one 4H-byte row as float4s,
// reductions go wave -> block -> one double atomic per block.
#include "mgv_common.h"
#include "mgv_slab.h"
#include "../../include/mgvae_hip.h"

namespace mgv {

__device__ __forceinline__ void atomic_add4(float* p, const float4& v) {
    atomicAdd(p + 0, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    // 64-lane shuffle reduction on the two 32-bit halves is not available for doubles: go through LDS
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    return red[0];
}

// ---------------------------------------------------------------------------------- decoder / recon
// one "item" = one edge handled by LPR lanes
template <int H, bool BWD>
__global__ __launch_bounds__(kThreads) void k_edge_dot(int64_t E, const float* s, const float* t, int ld,
                                                       const int64_t* src, const int64_t* dst, int sigmoid,
                                                       float* out, const float* gout, float* ds, float* dt) {
    constexpr int LPR = H / 4, EPB = kThreads / LPR;
    const int lr = threadIdx.x % LPR, slot = threadIdx.x / LPR;
    for (int64_t e0 = (int64_t)blockIdx.x * EPB; e0 < E; e0 += (int64_t)gridDim.x * EPB) {
        const int64_t e = e0 + slot;
        const bool ok = e < E;
        float4 a = zero4(), b = zero4();
        int64_t u = 0, v = 0;
        if (ok) { u = src[e]; v = dst[e]; a = ld4(s + u * ld + 4 * lr); b = ld4(t + v * ld + 4 * lr); }
        const float val = group_sum<LPR>(dot4(a, b));
        if (!BWD) {
            if (ok && lr == 0) out[e] = sigmoid ? sigmoidf_(val) : val;
        } else if (ok) {
            float c = gout[e];
            if (sigmoid) { const float p = sigmoidf_(val); c *= p * (1.0f - p); }
            atomic_add4(ds + u * ld + 4 * lr, scale4(c, b));
            atomic_add4(dt + v * ld + 4 * lr, scale4(c, a));
        }
    }
}

struct ReconArgs {
    const float* s; const float* t; int ld;
    const int64_t* psrc; const int64_t* pdst; int64_t Ep;
    const int64_t* nsrc; const int64_t* ndst; int64_t En;
    double* sums;               // [2]: sum -log(p+eps) over positives, sum -log(1-p+eps) over negatives
    unsigned long long* cnt;    // [4]: TP, FP, TN, FN
    int32_t* pred_bin;          // [Ep+En] or null
    const float* gscale;        // bwd: device scalar, upstream gradient of the loss
    float* ds; float* dt;       // bwd accumulators (same ld)
    double* slab;               // fwd: [gridDim][2] per-workgroup loss sums
};

template <int H, bool BWD>
__global__ __launch_bounds__(kThreads) void k_recon(ReconArgs a) {
    constexpr int LPR = H / 4, EPB = kThreads / LPR;
    __shared__ double red[kThreads];
    const int lr = threadIdx.x % LPR, slot = threadIdx.x / LPR;
    const int64_t E = a.Ep + a.En;
    float lp = 0.f, ln = 0.f;
    unsigned int tp = 0, fp = 0, tn = 0, fn = 0;
    float g = 0.f;
    if (BWD) g = *a.gscale;
    const float wpos = BWD ? g / (float)a.Ep : 0.f, wneg = BWD ? g / (float)a.En : 0.f;
    for (int64_t e0 = (int64_t)blockIdx.x * EPB; e0 < E; e0 += (int64_t)gridDim.x * EPB) {
        const int64_t e = e0 + slot;
        const bool ok = e < E;
        const bool pos = e < a.Ep;
        float4 x = zero4(), y = zero4();
        int64_t u = 0, v = 0;
        if (ok) {
            u = pos ? a.psrc[e] : a.nsrc[e - a.Ep];
            v = pos ? a.pdst[e] : a.ndst[e - a.Ep];
            x = ld4(a.s + u * a.ld + 4 * lr);
            y = ld4(a.t + v * a.ld + 4 * lr);
        }
        const float val = group_sum<LPR>(dot4(x, y));
        const float p = sigmoidf_(val);
        if (!BWD) {
            if (ok && lr == 0) {
                const bool hit = p > 0.5f;
                if (pos) { lp -= logf(p + 1e-15f); tp += hit; fn += !hit; }
                else     { ln -= logf((1.0f - p) + 1e-15f); fp += hit; tn += !hit; }
                if (a.pred_bin) a.pred_bin[e] = hit ? 1 : 0;
            }
        } else if (ok) {
            // d/dval of -log(p+eps) = -p(1-p)/(p+eps);  of -log(1-p+eps) = p(1-p)/(1-p+eps)
            const float dp = p * (1.0f - p);
            const float c = pos ? -wpos * dp / (p + 1e-15f) : wneg * dp / ((1.0f - p) + 1e-15f);
            atomic_add4(a.ds + u * a.ld + 4 * lr, scale4(c, y));
            atomic_add4(a.dt + v * a.ld + 4 * lr, scale4(c, x));
        }
    }
    if (!BWD) {
        const double sp = block_sum_d((double)lp, red);
        const double sn = block_sum_d((double)ln, red);
        const double c0 = block_sum_d((double)tp, red), c1 = block_sum_d((double)fp, red);
        const double c2 = block_sum_d((double)tn, red), c3 = block_sum_d((double)fn, red);
        if (threadIdx.x == 0) {
            a.slab[2 * blockIdx.x + 0] = sp; a.slab[2 * blockIdx.x + 1] = sn;      // this workgroup's row; added in a fixed order afterwards
            atomicAdd(a.cnt + 0, (unsigned long long)c0); atomicAdd(a.cnt + 1, (unsigned long long)c1);
            atomicAdd(a.cnt + 2, (unsigned long long)c2); atomicAdd(a.cnt + 3, (unsigned long long)c3);
        }
    }
}

// Backward of the reconstruction loss, node-centric over the batch graph's two CSRs: the positive
// edges ARE the graph edges, so  ds[u] = sum_{v in out(u)} c(u,v) t[v]  and  dt[v] = sum_{u in in(v)}
// c(u,v) s[u]  are gathers (each edge's score is recomputed from both ends); no atomics, every output
// row is written once, whole.
template <int H>
__global__ __launch_bounds__(kThreads) void k_recon_bwd_pull(int64_t N, const float* s, const float* t, int ld,
                                                             const int32_t* out_ptr, const int32_t* out_dst,
                                                             const int32_t* in_ptr, const int32_t* in_src, int64_t Ep,
                                                             const float* gscale, float* ds, float* dt) {
    constexpr int LPR = H / 4, RPB = kThreads / LPR;
    const int lr = threadIdx.x % LPR, slot = threadIdx.x / LPR;
    const float w = -(*gscale) / (float)Ep;
    for (int64_t n0 = (int64_t)blockIdx.x * RPB; n0 < N; n0 += (int64_t)gridDim.x * RPB) {
        const int64_t u = n0 + slot;
        if (u >= N) continue;
        const float4 su = ld4(s + u * ld + 4 * lr), tu = ld4(t + u * ld + 4 * lr);
        float4 gs = zero4(), gt = zero4();
        for (int e = out_ptr[u]; e < out_ptr[u + 1]; ++e) {          // u as a source
            const float4 tv = ld4(t + (int64_t)out_dst[e] * ld + 4 * lr);
            const float p = sigmoidf_(group_sum<LPR>(dot4(su, tv)));
            gs = fma4(w * p * (1.0f - p) / (p + 1e-15f), tv, gs);
        }
        for (int e = in_ptr[u]; e < in_ptr[u + 1]; ++e) {            // u as a destination
            const float4 sv = ld4(s + (int64_t)in_src[e] * ld + 4 * lr);
            const float p = sigmoidf_(group_sum<LPR>(dot4(sv, tu)));
            gt = fma4(w * p * (1.0f - p) / (p + 1e-15f), sv, gt);
        }
        float* pds = ds + u * ld + 4 * lr;
        float* pdt = dt + u * ld + 4 * lr;
        st4(pds, add4(ld4(pds), gs));
        st4(pdt, add4(ld4(pdt), gt));
    }
}

// Both halves of the loss from CSRs (positives = the batch graph, negatives bucketed by mgv_neg_bucket): every
// output row is WRITTEN once, no zero fill, no atomics.
//   * A node's four lists (positive out / in, negative out / in) are walked as ONE sequence in chunks of four partner rows: the
//     four row loads of a chunk are in flight together and the next chunk's partner ids are read while they fly (one dependent
//     memory round trip per four partners instead of one per partner: the lists average 1.6-2.6 entries each, 8 per node).
//   * Workgroups are dealt round-robin over the 8 XCDs (observed placement, speed only): XCD x takes the x-th CONTIGUOUS eighth of
//     the nodes, so the partners of the positive lists (neighbours in the same circuit, nearby ids) are served by the XCD's own L2.
template <int H>
__global__ __launch_bounds__(kThreads) void k_recon_bwd_pull2(int64_t N, const float* s, const float* t, int ld,
                                                              const int32_t* pout_ptr, const int32_t* pout_dst, const int32_t* pin_ptr,
                                                              const int32_t* pin_src, int64_t Ep, const int32_t* nout_ptr,
                                                              const int32_t* nout_dst, const int32_t* nin_ptr, const int32_t* nin_src,
                                                              int64_t En, const float* gscale, float* ds, float* dt, int skip_len) {
    constexpr int LPR = H / 4, RPB = kThreads / LPR, K = 4;
    const int lr = threadIdx.x % LPR, slot = threadIdx.x / LPR;
    const float wp = Ep > 0 ? -(*gscale) / (float)Ep : 0.f, wn = En > 0 ? (*gscale) / (float)En : 0.f;
    // node range of this workgroup: XCD-contiguous when the grid is a multiple of 8
    int64_t first, end, stride;
    if ((gridDim.x & 7) == 0) {
        const int64_t chunk = ((N + 8 * RPB - 1) / (8 * RPB)) * RPB;      // nodes per XCD, a multiple of the block's rows
        first = (blockIdx.x & 7) * chunk + (int64_t)(blockIdx.x >> 3) * RPB;
        end = min(N, ((blockIdx.x & 7) + 1) * chunk);
        stride = (int64_t)(gridDim.x >> 3) * RPB;
    } else {
        first = (int64_t)blockIdx.x * RPB; end = N; stride = (int64_t)gridDim.x * RPB;
    }
    for (int64_t n0 = first; n0 < end; n0 += stride) {
        const int64_t u = n0 + slot;
        if (u >= end) continue;
        const float4 su = ld4(s + u * ld + 4 * lr), tu = ld4(t + u * ld + 4 * lr);
        // the four lists as one sequence: [0, c0) positive out, [c0, c1) positive in, [c1, c2) negative out, [c2, c3) negative in
        int b0 = 0, b1 = 0, b2 = 0, b3 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (Ep > 0) {
            b0 = pout_ptr[u]; c0 = pout_ptr[u + 1] - b0;
            b1 = pin_ptr[u]; c1 = pin_ptr[u + 1] - b1;
            if (skip_len > 0 && c0 > skip_len) c0 = 0;       // a heavy list: left to mgv_recon_heavy_lists (one workgroup per segment)
            if (skip_len > 0 && c1 > skip_len) c1 = 0;
        }
        if (En > 0) {
            b2 = nout_ptr[u]; c2 = nout_ptr[u + 1] - b2;
            b3 = nin_ptr[u]; c3 = nin_ptr[u + 1] - b3;
        }
        c1 += c0; c2 += c1; c3 += c2;
        auto partner = [&](int j) -> int {                  // row id of sequence entry j (j < c3)
            return j < c0 ? pout_dst[b0 + j] : j < c1 ? pin_src[b1 + (j - c0)] : j < c2 ? nout_dst[b2 + (j - c1)] : nin_src[b3 + (j - c2)];
        };
        float4 gs = zero4(), gt = zero4();
        int id[K];
#pragma unroll
        for (int k = 0; k < K; ++k) id[k] = k < c3 ? partner(k) : 0;
        for (int j0 = 0; j0 < c3; j0 += K) {
            float4 v[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = j0 + k;
                const bool from_t = j < c0 || (j >= c1 && j < c2);          // out lists pair s[u] with t[partner]
                if (j < c3) v[k] = ld4((from_t ? t : s) + (int64_t)id[k] * ld + 4 * lr);
            }
            int nid[K];
#pragma unroll
            for (int k = 0; k < K; ++k) nid[k] = j0 + K + k < c3 ? partner(j0 + K + k) : 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = j0 + k;
                if (j < c3) {
                    const bool out = j < c0 || (j >= c1 && j < c2), pos = j < c1;
                    const float p = sigmoidf_(group_sum<LPR>(dot4(out ? su : tu, v[k])));
                    const float c = pos ? wp * p * (1.0f - p) / (p + 1e-15f) : wn * p * (1.0f - p) / ((1.0f - p) + 1e-15f);
                    if (out) gs = fma4(c, v[k], gs); else gt = fma4(c, v[k], gt);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) id[k] = nid[k];
        }
        st4(ds + u * ld + 4 * lr, gs);
        st4(dt + u * ld + 4 * lr, gt);
    }
}

// One segment of a heavy positive list per workgroup: sum over the segment's partners p of c(own, p) * row(p), c from the own row's
// dot product with the partner (which = 0: own s[u], partners t[.], result for ds[u]; which = 1: own t[u], partners s[.], for dt[u]).
// The 16 lane groups take partners strided, their partial sums meet in LDS in group order: deterministic.
template <int H>
__global__ __launch_bounds__(256) void k_recon_heavy_seg(int S, const int32_t* seg_node, const int32_t* seg_e0, const int32_t* seg_e1,
                                                         const int32_t* nodes, const float* s, const float* t, int ld, const int32_t* list,
                                                         int which, int64_t Ep, const float* gscale, float* partial) {
    constexpr int LPR = H / 4, G = 256 / LPR;
    __shared__ __attribute__((aligned(16))) float s_p[G][H];
    const int lr = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const float wp = -(*gscale) / (float)Ep;
    for (int sg = blockIdx.x; sg < S; sg += gridDim.x) {
        const int64_t u = nodes[seg_node[sg]];
        const float4 own = ld4((which ? t : s) + u * ld + 4 * lr);
        const float* other = which ? s : t;
        float4 acc = zero4();
        for (int e = seg_e0[sg] + grp; e < seg_e1[sg]; e += G) {
            const float4 v = ld4(other + (int64_t)list[e] * ld + 4 * lr);
            const float p = sigmoidf_(group_sum<LPR>(dot4(own, v)));
            acc = fma4(wp * p * (1.0f - p) / (p + 1e-15f), v, acc);
        }
        __syncthreads();
        st4(&s_p[grp][4 * lr], acc);
        __syncthreads();
        if (grp == 0) {
            float4 tot = zero4();
            for (int g = 0; g < G; ++g) tot = add4(tot, ld4(&s_p[g][4 * lr]));
            st4(partial + (int64_t)sg * H + 4 * lr, tot);
        }
    }
}

// Arbitrary edge lists (the sampled negatives): one edge per H-lane group, one float per lane, so every
// atomic wave-instruction adds whole contiguous rows (the shape the memory-side atomic units like).
template <int H>
__global__ __launch_bounds__(kThreads) void k_recon_bwd_rows(int64_t E, const float* s, const float* t, int ld, const int64_t* src,
                                                             const int64_t* dst, int negative, float wscale, const float* gscale,
                                                             float* ds, float* dt) {
    constexpr int EPB = kThreads / H;
    const int l = threadIdx.x % H, slot = threadIdx.x / H;
    const float w = (*gscale) * wscale;
    for (int64_t e0 = (int64_t)blockIdx.x * EPB; e0 < E; e0 += (int64_t)gridDim.x * EPB) {
        const int64_t e = e0 + slot;
        const bool ok = e < E;
        int64_t u = 0, v = 0;
        float x = 0.f, y = 0.f;
        if (ok) { u = src[e]; v = dst[e]; x = s[u * ld + l]; y = t[v * ld + l]; }
        const float p = sigmoidf_(group_sum<H>(x * y));
        if (!ok) continue;
        const float dp = p * (1.0f - p);
        const float c = negative ? w * dp / ((1.0f - p) + 1e-15f) : -w * dp / (p + 1e-15f);
        atomicAdd(ds + u * ld + l, c * y);
        atomicAdd(dt + v * ld + l, c * x);
    }
}

// ---------------------------------------------------------------------------------- functional loss
// ws (double[8]): 0 sum d, 1 sum d^2, 2 sum t, 3 sum t^2, 4 sum |zd - zt|, 5 sum sgn, 6 sum sgn*zd
template <int H>
__global__ __launch_bounds__(kThreads) void k_func_dist(int64_t P, const float* hf, const int64_t* pa, const int64_t* pb,
                                                        const float* tt, float eps, float* dis, double* slab) {      // slab [gridDim][4]
    constexpr int LPR = H / 4, PPB = kThreads / LPR;
    __shared__ double red[kThreads];
    const int lr = threadIdx.x % LPR, slot = threadIdx.x / LPR;
    double sd = 0, sd2 = 0, st = 0, st2 = 0;
    for (int64_t p0 = (int64_t)blockIdx.x * PPB; p0 < P; p0 += (int64_t)gridDim.x * PPB) {
        const int64_t p = p0 + slot;
        const bool ok = p < P;
        float4 x = zero4(), y = zero4();
        if (ok) { x = ld4(hf + pa[p] * H + 4 * lr); y = ld4(hf + pb[p] * H + 4 * lr); }
        const float xy = group_sum<LPR>(dot4(x, y));
        const float xx = group_sum<LPR>(dot4(x, x));
        const float yy = group_sum<LPR>(dot4(y, y));
        if (ok && lr == 0) {
            const float nx = fmaxf(sqrtf(xx), eps), ny = fmaxf(sqrtf(yy), eps);
            const float d = 1.0f - xy / (nx * ny);
            dis[p] = d;
            const float t = tt[p];
            sd += d; sd2 += (double)d * d; st += t; st2 += (double)t * t;
        }
    }
    sd = block_sum_d(sd, red); sd2 = block_sum_d(sd2, red); st = block_sum_d(st, red); st2 = block_sum_d(st2, red);
    if (threadIdx.x == 0) { double* row = slab + 4 * (int64_t)blockIdx.x; row[0] = sd; row[1] = sd2; row[2] = st; row[3] = st2; }
}

struct ZStats { float mu_d, inv_sd, mu_t, inv_st; };
__device__ __forceinline__ ZStats zstats(const double* ws, int64_t P) {
    // torch.std: unbiased (divide by P-1), utils/utils.py:32-36
    const double n = (double)P;
    const double md = ws[0] / n, mt = ws[2] / n;
    const double vd = (ws[1] - n * md * md) / (n - 1.0), vt = (ws[3] - n * mt * mt) / (n - 1.0);
    ZStats z;
    z.mu_d = (float)md; z.mu_t = (float)mt;
    z.inv_sd = (float)(1.0 / sqrt(vd > 0 ? vd : 0.0));
    z.inv_st = (float)(1.0 / sqrt(vt > 0 ? vt : 0.0));
    return z;
}

__global__ __launch_bounds__(kThreads) void k_func_l1(int64_t P, const float* dis, const float* tt, const double* ws, double* slab) {      // slab [gridDim][3]
    __shared__ double red[kThreads];
    const ZStats z = zstats(ws, P);
    double sl = 0, ss = 0, ssz = 0;
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < P; p += (int64_t)gridDim.x * kThreads) {
        const float zd = (dis[p] - z.mu_d) * z.inv_sd, zt = (tt[p] - z.mu_t) * z.inv_st;
        const float diff = zd - zt;
        const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        sl += fabsf(diff); ss += sg; ssz += (double)sg * zd;
    }
    sl = block_sum_d(sl, red); ss = block_sum_d(ss, red); ssz = block_sum_d(ssz, red);
    if (threadIdx.x == 0) { double* row = slab + 3 * (int64_t)blockIdx.x; row[0] = sl; row[1] = ss; row[2] = ssz; }
}

// dL/dd_q = g/(P sd) [sgn_q - mean(sgn) - zd_q * sum(sgn*zd)/(P-1)];  d = 1 - cos
template <int H>
__global__ __launch_bounds__(kThreads) void k_func_bwd(int64_t P, const float* hf, const int64_t* pa, const int64_t* pb,
                                                       const float* tt, const float* dis, float eps, const double* ws,
                                                       const float* gscale, float* dhf) {
    // one pair per H-lane group, one float per lane: every atomic wave-instruction adds whole contiguous rows (the shape
    // the memory-side atomic units take at full rate; a float4-per-lane layout issues four strided adds per row)
    constexpr int PPB = kThreads / H;
    const int l = threadIdx.x % H, slot = threadIdx.x / H;
    const ZStats z = zstats(ws, P);
    const float g = *gscale;
    const float mean_s = (float)(ws[5] / (double)P), ssz = (float)(ws[6] / ((double)P - 1.0));
    const float k = g / (float)P * z.inv_sd;
    for (int64_t p0 = (int64_t)blockIdx.x * PPB; p0 < P; p0 += (int64_t)gridDim.x * PPB) {
        const int64_t p = p0 + slot;
        const bool ok = p < P;
        float x = 0.f, y = 0.f;
        int64_t ia = 0, ib = 0;
        if (ok) { ia = pa[p]; ib = pb[p]; x = hf[ia * H + l]; y = hf[ib * H + l]; }
        const float xy = group_sum<H>(x * y);
        const float xx = group_sum<H>(x * x);
        const float yy = group_sum<H>(y * y);
        if (!ok) continue;
        const float zd = (dis[p] - z.mu_d) * z.inv_sd, zt = (tt[p] - z.mu_t) * z.inv_st;
        const float diff = zd - zt;
        const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        const float dd = k * (sg - mean_s - zd * ssz);      // dL/d dis
        const float dc = -dd;                               // dL/d cos
        const float rx = sqrtf(xx), ry = sqrtf(yy);
        const float nx = fmaxf(rx, eps), ny = fmaxf(ry, eps);
        const float inv = 1.0f / (nx * ny);
        const float cs = xy * inv;
        // cos = <x,y>/(nx ny); the norm factor only depends on x where it is not clamped
        const float kx = rx > eps ? cs / (nx * nx) : 0.f, ky = ry > eps ? cs / (ny * ny) : 0.f;
        atomicAdd(dhf + ia * H + l, dc * (y * inv - kx * x));
        atomicAdd(dhf + ib * H + l, dc * (x * inv - ky * y));
    }
}

// The same gradient pulled per node over the pairs it belongs to (lists grouped by first / by second member): every output row is
// written exactly once, in a fixed order — no zero fill in front, no float atomics, bit-reproducible.  A pair is visited twice
// (once from each member), which costs one extra gather of its other row.
template <int H>
__global__ __launch_bounds__(kThreads) void k_func_bwd_pull(int64_t N, int64_t P, const float* hf, const int64_t* pa, const int64_t* pb,
                                                            const float* tt, const float* dis, float eps, const double* ws, const float* gscale,
                                                            const int32_t* a_ptr, const int32_t* a_pair, const int32_t* b_ptr,
                                                            const int32_t* b_pair, const float* add, float* dhf) {
    constexpr int LPR = H / 4, RPB = kThreads / LPR;
    const int lr = threadIdx.x % LPR;
    const ZStats z = zstats(ws, P);
    const float g = *gscale;
    const float mean_s = (float)(ws[5] / (double)P), ssz = (float)(ws[6] / ((double)P - 1.0));
    const float k = g / (float)P * z.inv_sd;
    for (int64_t v = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR; v < N; v += (int64_t)gridDim.x * RPB) {
        const float4 own = ld4(hf + v * H + 4 * lr);
        const float oo = group_sum<LPR>(dot4(own, own));
        const float ro = sqrtf(oo), no = fmaxf(ro, eps);
        float4 acc = add ? ld4(add + v * H + 4 * lr) : zero4();      // gradient the same rows receive from another consumer
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            const int32_t* lp = side ? b_ptr : a_ptr;
            const int32_t* lq = side ? b_pair : a_pair;
            for (int e = lp[v]; e < lp[v + 1]; ++e) {
                const int p = lq[e];
                const int64_t u = side ? pa[p] : pb[p];
                const float4 oth = ld4(hf + u * H + 4 * lr);
                const float xy = group_sum<LPR>(dot4(own, oth));
                const float uu = group_sum<LPR>(dot4(oth, oth));
                const float zd = (dis[p] - z.mu_d) * z.inv_sd, zt = (tt[p] - z.mu_t) * z.inv_st;
                const float diff = zd - zt;
                const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                const float dc = -k * (sg - mean_s - zd * ssz);              // dL/d cos
                const float nu = fmaxf(sqrtf(uu), eps);
                const float inv = 1.0f / (no * nu);
                const float ko = ro > eps ? xy * inv / (no * no) : 0.f;      // the norm factor only depends on the own row where it is not clamped
                acc = fma4(dc * inv, oth, acc);
                acc = fma4(-dc * ko, own, acc);
            }
        }
        st4(dhf + v * H + 4 * lr, acc);
    }
}

// ---------------------------------------------------------------------------------- sampler + KL
__device__ __forceinline__ uint32_t mix32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}
__device__ __forceinline__ float gauss_from_counter(uint64_t seed, uint64_t i) {
    const uint32_t a = mix32(seed * 0x9E3779B97F4A7C15ULL + 2 * i), b = mix32(seed * 0x9E3779B97F4A7C15ULL + 2 * i + 1);
    const float u1 = ((a >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (b >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * __cosf(6.283185307179586f * u2);
}

// z = mu + exp(logstd) * eps;  klsum += sum(1 + 2 logstd - mu^2 - exp(logstd)^2)
__global__ __launch_bounds__(kThreads) void k_reparam_fwd(int64_t n, const float* mu, const float* ls, const float* eps,
                                                          uint64_t seed, float* eps_out, float* z, double* klsum) {
    __shared__ double red[kThreads];
    double acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float m = mu[i], l = ls[i];
        const float e = eps ? eps[i] : gauss_from_counter(seed, (uint64_t)i);
        if (eps_out) eps_out[i] = e;
        const float sd = expf(l);
        z[i] = m + sd * e;
        acc += (double)(1.0f + 2.0f * l - m * m - sd * sd);
    }
    acc = block_sum_d(acc, red);
    if (threadIdx.x == 0) atomicAdd(klsum, acc);
}

// dmu = gz + gkl * klcoef * (-2 mu);  dls = gz * eps * exp(ls) + gkl * klcoef * (2 - 2 exp(2 ls))
__global__ __launch_bounds__(kThreads) void k_reparam_bwd(int64_t n, const float* mu, const float* ls, const float* eps,
                                                          const float* gz, const float* gkl, float klcoef, float* dmu, float* dls) {
    const float gk = gkl ? (*gkl) * klcoef : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float m = mu[i], l = ls[i], sd = expf(l);
        const float g = gz ? gz[i] : 0.f;
        dmu[i] = g + gk * (-2.0f * m);
        dls[i] = g * eps[i] * sd + gk * (2.0f - 2.0f * sd * sd);
    }
}

__global__ __launch_bounds__(kThreads) void k_confusion(int64_t n, const int32_t* pred, const int32_t* gt, unsigned long long* cnt) {
    __shared__ double red[kThreads];
    unsigned int tp = 0, fp = 0, tn = 0, fn = 0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const int p = pred[i], g = gt[i];
        tp += (p == 1) & (g == 1); fp += (p == 1) & (g == 0); tn += (p == 0) & (g == 0); fn += (p == 0) & (g == 1);
    }
    const double c0 = block_sum_d((double)tp, red), c1 = block_sum_d((double)fp, red);
    const double c2 = block_sum_d((double)tn, red), c3 = block_sum_d((double)fn, red);
    if (threadIdx.x == 0) {
        atomicAdd(cnt + 0, (unsigned long long)c0); atomicAdd(cnt + 1, (unsigned long long)c1);
        atomicAdd(cnt + 2, (unsigned long long)c2); atomicAdd(cnt + 3, (unsigned long long)c3);
    }
}

inline int items_grid(int64_t items, int per_block) { return grid_for((items + per_block - 1) / per_block, 8); }

}  // namespace mgv

#define MGV_DISPATCH_H(H, CALL)                      \
    switch (H) {                                     \
        case 16: { constexpr int HH = 16; CALL; } break;  \
        case 32: { constexpr int HH = 32; CALL; } break;  \
        case 64: { constexpr int HH = 64; CALL; } break;  \
        case 128: { constexpr int HH = 128; CALL; } break; \
        default: return MGV_EUNSUPPORTED;            \
    }

extern "C" int mgv_edge_dot_fwd(int H, int64_t E, const float* s, const float* t, int ld, const int64_t* src,
                                const int64_t* dst, int sigmoid, float* out, void* stream) {
    MGV_CHECK_ARG(E >= 0 && s && t && out && ld >= H && ld % 4 == 0);
    if (E == 0) return MGV_OK;
    MGV_CHECK_ARG(src && dst);
    hipStream_t st = static_cast<hipStream_t>(stream);
    MGV_DISPATCH_H(H, hipLaunchKernelGGL((mgv::k_edge_dot<HH, false>), dim3(mgv::items_grid(E, mgv::kThreads / (HH / 4))), dim3(mgv::kThreads), 0, st,
                                         E, s, t, ld, src, dst, sigmoid, out, nullptr, nullptr, nullptr));
    MGV_LAUNCH_RET();
}

extern "C" int mgv_edge_dot_bwd(int H, int64_t E, const float* s, const float* t, int ld, const int64_t* src,
                                const int64_t* dst, int sigmoid, const float* gout, float* ds, float* dt, void* stream) {
    MGV_CHECK_ARG(E >= 0 && s && t && gout && ds && dt && ld >= H && ld % 4 == 0);
    if (E == 0) return MGV_OK;
    MGV_CHECK_ARG(src && dst);
    hipStream_t st = static_cast<hipStream_t>(stream);
    MGV_DISPATCH_H(H, hipLaunchKernelGGL((mgv::k_edge_dot<HH, true>), dim3(mgv::items_grid(E, mgv::kThreads / (HH / 4))), dim3(mgv::kThreads), 0, st,
                                         E, s, t, ld, src, dst, sigmoid, nullptr, gout, ds, dt));
    MGV_LAUNCH_RET();
}

extern "C" int mgv_recon_loss_fwd(int H, const float* s, const float* t, int ld, const int64_t* pos_src, const int64_t* pos_dst,
                                  int64_t Epos, const int64_t* neg_src, const int64_t* neg_dst, int64_t Eneg,
                                  double* sums, uint64_t* counts, int32_t* pred_bin, double* workspace, int64_t workspace_doubles,
                                  void* stream) {
    MGV_CHECK_ARG(s && t && sums && counts && Epos >= 0 && Eneg >= 0 && ld >= H && ld % 4 == 0);
    MGV_CHECK_ARG((Epos == 0 || (pos_src && pos_dst)) && (Eneg == 0 || (neg_src && neg_dst)));
    if (Epos + Eneg == 0) return MGV_OK;
    mgv::ReconArgs a{};
    a.s = s; a.t = t; a.ld = ld; a.psrc = pos_src; a.pdst = pos_dst; a.Ep = Epos; a.nsrc = neg_src; a.ndst = neg_dst; a.En = Eneg;
    a.sums = sums; a.cnt = reinterpret_cast<unsigned long long*>(counts); a.pred_bin = pred_bin;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int grid = 0;
    MGV_DISPATCH_H(H, grid = mgv::items_grid(Epos + Eneg, mgv::kThreads / (HH / 4)));
    MGV_CHECK_ARG(workspace && workspace_doubles >= 2 * (int64_t)grid);
    a.slab = workspace;
    MGV_DISPATCH_H(H, hipLaunchKernelGGL((mgv::k_recon<HH, false>), dim3(grid), dim3(mgv::kThreads), 0, st, a));
    mgv::launch_slab_sum<double, double>(workspace, grid, 2, 2, sums, st);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_recon_loss_bwd(int H, int64_t N, const float* s, const float* t, int ld, const int64_t* pos_src, const int64_t* pos_dst,
                                  int64_t Epos, const int32_t* pos_out_ptr, const int32_t* pos_out_dst, const int32_t* pos_in_ptr,
                                  const int32_t* pos_in_src, const int64_t* neg_src, const int64_t* neg_dst, int64_t Eneg,
                                  const float* gscale, float* ds, float* dt, void* stream) {
    MGV_CHECK_ARG(s && t && gscale && ds && dt && Epos >= 0 && Eneg >= 0 && N >= 0 && ld >= H && ld % 4 == 0);
    MGV_CHECK_ARG((Epos == 0 || (pos_src && pos_dst)) && (Eneg == 0 || (neg_src && neg_dst)));
    const bool pull = pos_out_ptr != nullptr;
    MGV_CHECK_ARG(!pull || (pos_in_ptr && (Epos == 0 || (pos_out_dst && pos_in_src))));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (Epos > 0 && pull) {
        MGV_DISPATCH_H(H, hipLaunchKernelGGL((mgv::k_recon_bwd_pull<HH>), dim3(mgv::items_grid(N, mgv::kThreads / (HH / 4))), dim3(mgv::kThreads), 0, st,
                                             N, s, t, ld, pos_out_ptr, pos_out_dst, pos_in_ptr, pos_in_src, Epos, gscale, ds, dt));
    }
    for (int neg = (Epos > 0 && pull) ? 1 : 0; neg < 2; ++neg) {
        const int64_t E = neg ? Eneg : Epos;
        if (E == 0) continue;
        const int64_t* es = neg ? neg_src : pos_src;
        const int64_t* ed = neg ? neg_dst : pos_dst;
        const float wsc = 1.0f / (float)E;
        switch (H) {     // one float per lane: the row must fit a wave
            case 16: hipLaunchKernelGGL((mgv::k_recon_bwd_rows<16>), dim3(mgv::items_grid(E, mgv::kThreads / 16)), dim3(mgv::kThreads), 0, st, E, s, t, ld, es, ed, neg, wsc, gscale, ds, dt); break;
            case 32: hipLaunchKernelGGL((mgv::k_recon_bwd_rows<32>), dim3(mgv::items_grid(E, mgv::kThreads / 32)), dim3(mgv::kThreads), 0, st, E, s, t, ld, es, ed, neg, wsc, gscale, ds, dt); break;
            case 64: hipLaunchKernelGGL((mgv::k_recon_bwd_rows<64>), dim3(mgv::items_grid(E, mgv::kThreads / 64)), dim3(mgv::kThreads), 0, st, E, s, t, ld, es, ed, neg, wsc, gscale, ds, dt); break;
            default: return MGV_EUNSUPPORTED;
        }
    }
    MGV_LAUNCH_RET();
}

extern "C" int mgv_recon_loss_bwd_csr(int H, int64_t N, const float* s, const float* t, int ld, const int32_t* pos_out_ptr,
                                      const int32_t* pos_out_dst, const int32_t* pos_in_ptr, const int32_t* pos_in_src, int64_t Epos,
                                      const int32_t* neg_out_ptr, const int32_t* neg_out_dst, const int32_t* neg_in_ptr,
                                      const int32_t* neg_in_src, int64_t Eneg, const float* gscale, float* ds, float* dt,
                                      int skip_pos_longer_than, void* stream) {
    MGV_CHECK_ARG(s && t && gscale && ds && dt && Epos >= 0 && Eneg >= 0 && N >= 0 && ld >= H && ld % 4 == 0 && skip_pos_longer_than >= 0);
    MGV_CHECK_ARG(Epos == 0 || (pos_out_ptr && pos_out_dst && pos_in_ptr && pos_in_src));
    MGV_CHECK_ARG(Eneg == 0 || (neg_out_ptr && neg_out_dst && neg_in_ptr && neg_in_src));
    if (N == 0) return MGV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MGV_DISPATCH_H(H, hipLaunchKernelGGL((mgv::k_recon_bwd_pull2<HH>), dim3(mgv::items_grid(N, mgv::kThreads / (HH / 4))), dim3(mgv::kThreads), 0, st,
                                         N, s, t, ld, pos_out_ptr, pos_out_dst, pos_in_ptr, pos_in_src, Epos, neg_out_ptr, neg_out_dst,
                                         neg_in_ptr, neg_in_src, Eneg, gscale, ds, dt, skip_pos_longer_than));
    MGV_LAUNCH_RET();
}

extern "C" int mgv_recon_heavy_lists(int H, const float* s, const float* t, int ld, int64_t Epos, const float* gscale, int K,
                                     const int32_t* nodes, const int32_t* node_seg_ptr, int S, const int32_t* seg_node, const int32_t* seg_e0,
                                     const int32_t* seg_e1, const int32_t* list, int which, float* partial_ws, float* out, void* stream) {
    MGV_CHECK_ARG(K >= 0 && S >= 0 && Epos > 0 && s && t && gscale && out && ld >= H && ld % 4 == 0 && (which == 0 || which == 1));
    if (K == 0 || S == 0) return MGV_OK;
    MGV_CHECK_ARG(nodes && node_seg_ptr && seg_node && seg_e0 && seg_e1 && list && partial_ws);
    hipStream_t st = static_cast<hipStream_t>(stream);
    MGV_DISPATCH_H(H, hipLaunchKernelGGL((mgv::k_recon_heavy_seg<HH>), dim3(S < 4096 ? S : 4096), dim3(256), 0, st, S, seg_node, seg_e0, seg_e1, nodes,
                                         s, t, ld, list, which, Epos, gscale, partial_ws));
    hipLaunchKernelGGL(mgv::k_heavy_add, dim3((K + 15) / 16 < 1024 ? (K + 15) / 16 : 1024), dim3(256), 0, st, K, H, nodes, node_seg_ptr, partial_ws, out, ld, 1);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_func_loss_fwd(int H, int64_t P, const float* hf, const int64_t* pair_a, const int64_t* pair_b,
                                 const float* tt, float eps, float* dis, double* ws, double* workspace, int64_t workspace_doubles,
                                 void* stream) {
    MGV_CHECK_ARG(P >= 2 && hf && pair_a && pair_b && tt && dis && ws);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int g1 = 0;
    MGV_DISPATCH_H(H, g1 = mgv::items_grid(P, mgv::kThreads / (HH / 4)));
    const int g2 = mgv::items_grid(P, mgv::kThreads);
    MGV_CHECK_ARG(workspace && workspace_doubles >= 4 * (int64_t)g1 && workspace_doubles >= 3 * (int64_t)g2);
    MGV_DISPATCH_H(H, hipLaunchKernelGGL((mgv::k_func_dist<HH>), dim3(g1), dim3(mgv::kThreads), 0, st, P, hf, pair_a, pair_b, tt, eps, dis, workspace));
    mgv::launch_slab_sum<double, double>(workspace, g1, 4, 4, ws, st);
    hipLaunchKernelGGL(mgv::k_func_l1, dim3(g2), dim3(mgv::kThreads), 0, st, P, dis, tt, ws, workspace);
    mgv::launch_slab_sum<double, double>(workspace, g2, 3, 3, ws + 4, st);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_func_loss_bwd_csr(int H, int64_t N, int64_t P, const float* hf, const int64_t* pair_a, const int64_t* pair_b, const float* tt,
                                     const float* dis, float eps, const double* ws, const float* gscale, const int32_t* a_ptr,
                                     const int32_t* a_pair, const int32_t* b_ptr, const int32_t* b_pair, const float* add, float* dhf, void* stream) {
    MGV_CHECK_ARG(N >= 0 && P >= 2 && hf && pair_a && pair_b && tt && dis && ws && gscale && a_ptr && a_pair && b_ptr && b_pair && dhf);
    if (N == 0) return MGV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (H) {
        case 16: hipLaunchKernelGGL((mgv::k_func_bwd_pull<16>), dim3(mgv::items_grid(N, mgv::kThreads / 4)), dim3(mgv::kThreads), 0, st, N, P, hf, pair_a, pair_b, tt, dis, eps, ws, gscale, a_ptr, a_pair, b_ptr, b_pair, add, dhf); break;
        case 32: hipLaunchKernelGGL((mgv::k_func_bwd_pull<32>), dim3(mgv::items_grid(N, mgv::kThreads / 8)), dim3(mgv::kThreads), 0, st, N, P, hf, pair_a, pair_b, tt, dis, eps, ws, gscale, a_ptr, a_pair, b_ptr, b_pair, add, dhf); break;
        case 64: hipLaunchKernelGGL((mgv::k_func_bwd_pull<64>), dim3(mgv::items_grid(N, mgv::kThreads / 16)), dim3(mgv::kThreads), 0, st, N, P, hf, pair_a, pair_b, tt, dis, eps, ws, gscale, a_ptr, a_pair, b_ptr, b_pair, add, dhf); break;
        default: return MGV_EUNSUPPORTED;
    }
    MGV_LAUNCH_RET();
}

extern "C" int mgv_func_loss_bwd(int H, int64_t P, const float* hf, const int64_t* pair_a, const int64_t* pair_b,
                                 const float* tt, const float* dis, float eps, const double* ws, const float* gscale,
                                 float* dhf, void* stream) {
    MGV_CHECK_ARG(P >= 2 && hf && pair_a && pair_b && tt && dis && ws && gscale && dhf);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (H) {     // one float per lane: the row must fit a wave
        case 16: hipLaunchKernelGGL((mgv::k_func_bwd<16>), dim3(mgv::items_grid(P, mgv::kThreads / 16)), dim3(mgv::kThreads), 0, st, P, hf, pair_a, pair_b, tt, dis, eps, ws, gscale, dhf); break;
        case 32: hipLaunchKernelGGL((mgv::k_func_bwd<32>), dim3(mgv::items_grid(P, mgv::kThreads / 32)), dim3(mgv::kThreads), 0, st, P, hf, pair_a, pair_b, tt, dis, eps, ws, gscale, dhf); break;
        case 64: hipLaunchKernelGGL((mgv::k_func_bwd<64>), dim3(mgv::items_grid(P, mgv::kThreads / 64)), dim3(mgv::kThreads), 0, st, P, hf, pair_a, pair_b, tt, dis, eps, ws, gscale, dhf); break;
        default: return MGV_EUNSUPPORTED;
    }
    MGV_LAUNCH_RET();
}

extern "C" int mgv_reparam_fwd(int64_t n, const float* mu, const float* logstd, const float* eps, uint64_t seed,
                               float* eps_out, float* z, double* klsum, void* stream) {
    MGV_CHECK_ARG(n >= 0 && mu && logstd && z && klsum);
    if (n == 0) return MGV_OK;
    hipLaunchKernelGGL(mgv::k_reparam_fwd, dim3(mgv::items_grid(n, mgv::kThreads)), dim3(mgv::kThreads), 0,
                       static_cast<hipStream_t>(stream), n, mu, logstd, eps, seed, eps_out, z, klsum);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_reparam_bwd(int64_t n, const float* mu, const float* logstd, const float* eps, const float* gz,
                               const float* gkl, float klcoef, float* dmu, float* dlogstd, void* stream) {
    MGV_CHECK_ARG(n >= 0 && mu && logstd && eps && dmu && dlogstd);
    if (n == 0) return MGV_OK;
    hipLaunchKernelGGL(mgv::k_reparam_bwd, dim3(mgv::items_grid(n, mgv::kThreads)), dim3(mgv::kThreads), 0,
                       static_cast<hipStream_t>(stream), n, mu, logstd, eps, gz, gkl, klcoef, dmu, dlogstd);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_confusion(int64_t n, const int32_t* pred_bin, const int32_t* gt_bin, uint64_t* counts, void* stream) {
    MGV_CHECK_ARG(n >= 0 && counts);
    if (n == 0) return MGV_OK;
    MGV_CHECK_ARG(pred_bin && gt_bin);
    hipLaunchKernelGGL(mgv::k_confusion, dim3(mgv::items_grid(n, mgv::kThreads)), dim3(mgv::kThreads), 0,
                       static_cast<hipStream_t>(stream), n, pred_bin, gt_bin, reinterpret_cast<unsigned long long*>(counts));
    MGV_LAUNCH_RET();
}
