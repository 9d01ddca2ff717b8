// On-device batch builder: edge list -> in/out CSR, ASAP levels, (level, gate-type) buckets and
// 64-node tiles, as HIP kernels.  Replaces, on the GPU, the torch sorts / scans of deepgate/graph_plan.py and the host
// frontier loop of deepgate/parser.py::forward_levels; results are IDENTICAL to those (same stable orders), which the
// GPU tests assert array by array.  What the reference does instead: `torch.stack([ei[1], ei[0]])` per stage
// (digae_layer.py:264), boolean level / gate masks per level (dg_ae_model_aig.py:72-75), per-node edge scans
// (utils/dag_utils.py:91-105) and the O(levels x E) numpy rounds of top_sort (utils/dag_utils.py:10-37).
//
// Building blocks (all int32, HBM-bound integer work; int atomics run at L2 speed, unlike float atomics):
//   * exclusive scan: 2,048 items per block (local scan + block total), one block over the totals, add-back;
//   * CSR: degree histogram (atomicAdd), scan, cursor fill of EDGE IDS, then every node sorts its (short) list of edge ids
//     ascending — the result is the stable order of a stable sort by destination / source — and a gather pass;
//   * stable counting sort by a small key (level * T + slot <= a few thousand): per-block LDS histograms laid out
//     [key][block], ONE flat exclusive scan gives every (key, block) its base, blocks place their nodes in wave order;
//   * ASAP levels: frontier relaxation over the out-CSR (Kahn), one launch per level, frontier sizes stay on the device.
#include "mgv_common.h"
#include "../../include/mgvae_hip.h"

namespace mgv {

constexpr int kScanItems = 2048;           // items per scan block (256 threads x 8)
constexpr int kSortBlock = 1024;           // items per counting-sort block
constexpr int kMaxKeys = 8192;             // LDS histogram bins of the counting sort

// ------------------------------------------------------------------------------------------------ scan
__global__ __launch_bounds__(256) void k_scan_block(int64_t n, const int32_t* in, int32_t* out, int32_t* sums) {
    __shared__ int s_w[4];
    const int64_t base = (int64_t)blockIdx.x * kScanItems + threadIdx.x * 8;
    int v[8], t = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = base + k < n ? in[base + k] : 0; t += v[k]; }
    // exclusive prefix of t over the 256 threads
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < w; ++k) woff += s_w[k];
    int run = woff + inc - t;
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255) sums[blockIdx.x] = run;
}

// one block: exclusive scan of the block totals in place, grand total appended at sums[nb]
__global__ __launch_bounds__(1024) void k_scan_sums(int nb, int32_t* sums) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const int t = i < nb ? sums[i] : 0;
        int inc = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int woff = s_carry;
        for (int k = 0; k < w; ++k) woff += s_w[k];
        if (i < nb) sums[i] = woff + inc - t;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[nb] = s_carry;
}

__global__ __launch_bounds__(256) void k_scan_add(int64_t n, int32_t* out, const int32_t* sums, int nb) {
    const int off = sums[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kScanItems + threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (base + k < n) out[base + k] += off;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = sums[nb];       // total behind the last element
}

// short inputs (most scans of the plan builder: per-colour tables, tile counts): ONE workgroup, one launch instead of three
__global__ __launch_bounds__(1024) void k_scan_small(int64_t n, const int32_t* in, int32_t* out) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t b0 = 0; b0 < n; b0 += 1024) {
        const int64_t i = b0 + threadIdx.x;
        const int t = i < n ? in[i] : 0;
        int inc = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int woff = s_carry;
        for (int k = 0; k < w; ++k) woff += s_w[k];
        if (i < n) out[i] = woff + inc - t;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = s_carry;
}

// out[0..n] = exclusive prefix sums of in[0..n), out[n] = total; scratch >= ceil(n / 2048) + 1 ints
static int scan_exclusive(int64_t n, const int32_t* in, int32_t* out, int32_t* scratch, hipStream_t st) {
    const int nb = (int)((n + kScanItems - 1) / kScanItems);
    if (n == 0) { hipMemsetAsync(out, 0, sizeof(int32_t), st); MGV_LAUNCH_RET(); }
    if (n <= 32768 && in != out) { hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, st, n, in, out); MGV_LAUNCH_RET(); }
    hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(256), 0, st, n, in, out, scratch);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, nb, scratch);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, st, n, out, scratch, nb);
    MGV_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------------ CSR
__global__ __launch_bounds__(256) void k_csr_count(int64_t E, int64_t N, const int64_t* src, const int64_t* dst, int32_t* deg_in, int32_t* deg_out,
                                                   int32_t* err) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
        const int64_t s = src[e], d = dst[e];
        if (s < 0 || s >= N || d < 0 || d >= N) { *err = 1; continue; }
        atomicAdd(deg_in + d, 1);
        atomicAdd(deg_out + s, 1);
    }
}

__global__ __launch_bounds__(256) void k_csr_fill(int64_t E, int64_t N, const int64_t* src, const int64_t* dst, int32_t* cur_in, int32_t* cur_out,
                                                  int32_t* eid_in, int32_t* eid_out) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
        const int64_t s = src[e], d = dst[e];
        if (s < 0 || s >= N || d < 0 || d >= N) continue;
        eid_in[atomicAdd(cur_in + d, 1)] = (int32_t)e;
        eid_out[atomicAdd(cur_out + s, 1)] = (int32_t)e;
    }
}

// every node's list of edge ids ascending (= original edge order).  Short lists in registers; long ones are handed to
// k_sort_heavy through a list.
constexpr int kShortList = 32;
__global__ __launch_bounds__(256) void k_sort_lists(int64_t N, const int32_t* ptr, int32_t* eid, int32_t* heavy, int32_t* n_heavy) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int p0 = ptr[n], d = ptr[n + 1] - p0;
    if (d <= 1) return;
    if (d > kShortList) { heavy[atomicAdd(n_heavy, 1)] = (int32_t)n; return; }
    int v[kShortList];
#pragma unroll
    for (int k = 0; k < kShortList; ++k) v[k] = k < d ? eid[p0 + k] : 0x7fffffff;
    // odd-even transposition network on a fixed 32-entry array: no data-dependent indexing (stays in registers)
#pragma unroll 1
    for (int pass = 0; pass < d; ++pass) {
#pragma unroll
        for (int k = 0; k + 1 < kShortList; k += 2) { const int a = min(v[k], v[k + 1]), b = max(v[k], v[k + 1]); v[k] = a; v[k + 1] = b; }
#pragma unroll
        for (int k = 1; k + 1 < kShortList; k += 2) { const int a = min(v[k], v[k + 1]), b = max(v[k], v[k + 1]); v[k] = a; v[k + 1] = b; }
    }
#pragma unroll
    for (int k = 0; k < kShortList; ++k) if (k < d) eid[p0 + k] = v[k];
}

// one block per long list: bitonic sort in LDS up to 4,096 entries, rank sort (O(d^2 / 1024), rare) beyond
constexpr int kHeavyLds = 4096;
__global__ __launch_bounds__(1024) void k_sort_heavy(const int32_t* heavy, const int32_t* n_heavy, const int32_t* ptr, int32_t* eid, int32_t* tmp) {
    __shared__ int s_v[kHeavyLds];
    for (int h = blockIdx.x; h < *n_heavy; h += gridDim.x) {
        const int n = heavy[h];
        const int p0 = ptr[n], d = ptr[n + 1] - p0;
        if (d <= kHeavyLds) {
            int m = 1;
            while (m < d) m <<= 1;
            for (int i = threadIdx.x; i < m; i += 1024) s_v[i] = i < d ? eid[p0 + i] : 0x7fffffff;
            __syncthreads();
            for (int k = 2; k <= m; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = threadIdx.x; i < m; i += 1024) {
                        const int l = i ^ j;
                        if (l > i) {
                            const int a = s_v[i], b = s_v[l];
                            const bool up = (i & k) == 0;
                            if ((a > b) == up) { s_v[i] = b; s_v[l] = a; }
                        }
                    }
                    __syncthreads();
                }
            for (int i = threadIdx.x; i < d; i += 1024) eid[p0 + i] = s_v[i];
            __syncthreads();
        } else {
            for (int i = threadIdx.x; i < d; i += 1024) {          // rank = number of smaller values (equal ones in index order)
                const int a = eid[p0 + i];
                int r = 0;
                for (int j = 0; j < d; ++j) { const int b = eid[p0 + j]; r += (b < a) || (b == a && j < i); }
                tmp[p0 + r] = a;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < d; i += 1024) eid[p0 + i] = tmp[p0 + i];
            __syncthreads();
        }
    }
}

// Edges with a node id outside [0, N) are skipped by the count / fill kernels (status[0] = 1), so only the first in_ptr[N] (= out_ptr[N])
// slots of eid_in / eid_out are filled: the finish kernels stop there and zero the tail of their outputs instead of following an
// uninitialised edge id (the caller raises on the status; nothing launched before it reads the status may fault).
__global__ __launch_bounds__(256) void k_csr_finish_in(int64_t E, int64_t N, const int32_t* in_ptr, const int64_t* src, const int64_t* dst,
                                                       const int32_t* eid_in, int32_t* in_src, int32_t* in_dst, int32_t* pos_in) {
    const int64_t valid = in_ptr[N];
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < E; s += (int64_t)gridDim.x * 256) {
        if (s >= valid) { in_src[s] = 0; in_dst[s] = 0; continue; }
        const int e = eid_in[s];
        in_src[s] = (int32_t)src[e];
        in_dst[s] = (int32_t)dst[e];
        pos_in[e] = (int32_t)s;
    }
}
__global__ __launch_bounds__(256) void k_csr_finish_out(int64_t E, int64_t N, const int32_t* out_ptr, const int64_t* dst, const int32_t* eid_out,
                                                        const int32_t* pos_in, int32_t* out_dst, int32_t* out_slot) {
    const int64_t valid = out_ptr[N];
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < E; t += (int64_t)gridDim.x * 256) {
        if (t >= valid) { out_dst[t] = 0; out_slot[t] = 0; continue; }
        const int e = eid_out[t];
        out_dst[t] = (int32_t)dst[e];
        out_slot[t] = pos_in[e];
    }
}

// ------------------------------------------------------------------------------------------------ levels (Kahn over the out-CSR)
__global__ __launch_bounds__(256) void k_level_seed(int64_t N, const int32_t* in_ptr, int32_t* pending, int32_t* level, int32_t* frontier, int32_t* fcount) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int d = in_ptr[n + 1] - in_ptr[n];
    pending[n] = d;
    level[n] = 0;
    if (d == 0) frontier[atomicAdd(fcount, 1)] = (int32_t)n;
}
// one level: children of the frontier lose one pending parent per edge; those that reach zero form the next frontier
__global__ __launch_bounds__(256) void k_level_step(const int32_t* frontier, const int32_t* fcount, int32_t* next, int32_t* ncount, const int32_t* out_ptr,
                                                    const int32_t* out_dst, int32_t* pending, int32_t* level, int cur, int32_t* done) {
    const int nf = *fcount;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nf; i += gridDim.x * 256) {
        const int n = frontier[i];
        for (int e = out_ptr[n]; e < out_ptr[n + 1]; ++e) {
            const int c = out_dst[e];
            if (atomicSub(pending + c, 1) == 1) { level[c] = cur + 1; next[atomicAdd(ncount, 1)] = c; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(done, nf);
}

// ------------------------------------------------------------------------------------------------ buckets
struct SlotTable { uint8_t slot[256]; };

__global__ __launch_bounds__(256) void k_bucket_keys(int64_t N, int T, const float* gate, const int64_t* level64, SlotTable tab, uint8_t* gslot,
                                                     int32_t* level32, int32_t* key, int32_t* maxlevel) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int lmax = 0;
    if (n < N) {
        int g = (int)gate[n];
        g = g < 0 ? 0 : (g > 255 ? 255 : g);
        const int s = tab.slot[g];
        const int64_t lv = level64[n];
        const bool active = lv >= 1 && s != 255;
        gslot[n] = active ? (uint8_t)s : (uint8_t)255;
        level32[n] = (int32_t)lv;
        key[n] = active ? (int32_t)(lv * T + s) : -1;
        lmax = (int32_t)lv;
    }
    // one atomic per workgroup, and only when it can still raise the maximum (atomics on ONE address serialise in L2 at ~10 ns each:
    // one per node 0.75 ms at config 2, one per wave 0.70 ms, this ~0.03 ms)
    __shared__ int s_max[4];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) lmax = max(lmax, __shfl_xor(lmax, m, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = lmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        lmax = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        if (lmax > __hip_atomic_load(maxlevel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxlevel, lmax);
    }
}

// every source of an updated node must sit on a strictly lower level
__global__ __launch_bounds__(256) void k_check_levels(int64_t E, const int32_t* in_src, const int32_t* in_dst, const uint8_t* gslot, const int32_t* level,
                                                      int32_t* err) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
        const int d = in_dst[e], s = in_src[e];
        if (gslot[d] != 255 && level[s] >= level[d]) *err = 2;
    }
}

// counting sort, step 1: cnt[key * B + block] = items of `key` in the block (items with key < 0 are skipped)
__global__ __launch_bounds__(kSortBlock) void k_sort_hist(int64_t n, const int32_t* key, int K, int B, int32_t* cnt) {
    __shared__ int s_h[kMaxKeys];
    for (int i = threadIdx.x; i < K; i += kSortBlock) s_h[i] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kSortBlock + threadIdx.x;
    const int k = i < n ? key[i] : -1;
    if (k >= 0) atomicAdd(s_h + k, 1);
    __syncthreads();
    for (int j = threadIdx.x; j < K; j += kSortBlock) if (s_h[j]) cnt[(int64_t)j * B + blockIdx.x] = s_h[j];
}
// step 3 (after the flat scan of cnt into base): place items; within a block wave by wave, within a wave by lane: stable
__global__ __launch_bounds__(kSortBlock) void k_sort_place(int64_t n, const int32_t* key, int K, int B, const int32_t* base, int32_t* order) {
    __shared__ int s_c[kMaxKeys];
    for (int i = threadIdx.x; i < K; i += kSortBlock) s_c[i] = base[(int64_t)i * B + blockIdx.x];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kSortBlock + threadIdx.x;
    const int k = i < n ? key[i] : -1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int turn = 0; turn < kSortBlock / 64; ++turn) {
        if (w == turn) {
            unsigned long long todo = __ballot(k >= 0);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int kk = __shfl(k, leader, 64);
                const unsigned long long same = __ballot(k == kk) & todo;
                if (k == kk) {
                    const int r = s_c[kk] + __popcll(same & ((1ull << lane) - 1ull));
                    order[r] = (int32_t)i;
                }
                if (lane == leader) s_c[kk] += __popcll(same);
                todo &= ~same;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_key_starts(int K, int B, const int32_t* base, int32_t* key_start) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k <= K) key_start[k] = base[(int64_t)k * B];      // k = K: the total behind the last (key, block) cell
}
__global__ __launch_bounds__(256) void k_key_counts(int K, int B, const int32_t* base, int32_t* ntile) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const int c = base[(int64_t)(k + 1) * B] - base[(int64_t)k * B];       // base has K*B + 1 entries
    ntile[k] = (c + kTileRows - 1) / kTileRows;
}
__global__ __launch_bounds__(256) void k_tiles_fill(int K, int B, int T, const int32_t* base, const int32_t* tile_first, int32_t* tile_start,
                                                    int32_t* tile_count, int32_t* tile_slot) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const int s0 = base[(int64_t)k * B], c = base[(int64_t)(k + 1) * B] - s0;
    for (int t = 0, f = tile_first[k]; t * kTileRows < c; ++t) {
        tile_start[f + t] = s0 + t * kTileRows;
        tile_count[f + t] = min(kTileRows, c - t * kTileRows);
        tile_slot[f + t] = k % T;
    }
}
__global__ __launch_bounds__(256) void k_order_span(int64_t n_active, const int32_t* order, const int32_t* in_ptr, const int32_t* out_ptr, int4* span) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_active) return;
    const int n = order[i];
    span[i] = make_int4(in_ptr[n], in_ptr[n + 1], out_ptr[n], out_ptr[n + 1]);
}
__global__ __launch_bounds__(256) void k_level_tile_ptr(int L, int T, const int32_t* tile_first, int32_t* ltp) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l <= L) ltp[l] = tile_first[min(l * T, L * T)];
}

// (degree, class) pairs of the forward CSR -> dense class ids (first half round of an encoder, digae_layer.py:260)
__global__ __launch_bounds__(256) void k_pair_mark(int64_t N, const int32_t* in_ptr, const uint8_t* xcls, int32_t* present, int32_t* err) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int d = in_ptr[n + 1] - in_ptr[n];
    if (d > 255) { *err = 3; return; }
    present[d * 256 + xcls[n]] = 1;
}
__global__ __launch_bounds__(256) void k_pair_ids(int64_t N, const int32_t* in_ptr, const uint8_t* xcls, const int32_t* rank, int32_t* cid) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int d = min(in_ptr[n + 1] - in_ptr[n], 255);
    cid[n] = rank[d * 256 + xcls[n]];
}
__global__ __launch_bounds__(256) void k_pair_table(const int32_t* present, const int32_t* rank, int32_t* cls_deg, uint8_t* cls_x) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < 65536 && present[p]) { cls_deg[rank[p]] = p >> 8; cls_x[rank[p]] = (uint8_t)(p & 255); }
}

static inline int blocks_for(int64_t n, int per = 256, int cap = 256 * 64) { const int64_t b = (n + per - 1) / per; return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

// ---- colour refinement (GraphPlan.quotient): a node's colour after a half round is (feature class, previous colour, multiset of its
// neighbours' previous colours).  k_colour_keys folds that into a 64-bit grouping key (two sums of random per-colour values: order
// independent); k_colour_check then compares every node with its group's representative EXACTLY — class, previous colour, degree and
// the two neighbour-colour multisets (every colour of the node's list must occur equally often in both lists) — so a key collision is
// found, never believed.  Lists beyond kColourCheckMax entries are left to the caller's sort-based check (n_long counts them).
constexpr int kColourCheckMax = 48;

__global__ __launch_bounds__(256) void k_colour_keys(int64_t N, const int32_t* ptr, const int32_t* idx, const int32_t* prev, const int64_t* f, int64_t fstride,
                                                     const uint8_t* xcls, int key_bits, int64_t* key) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int e0 = ptr[i], e1 = ptr[i + 1];
    unsigned long long h1 = 0, h2 = 0;
    for (int e = e0; e < e1; ++e) {
        const int c = prev[idx[e]];
        h1 += (unsigned long long)f[c];
        h2 += (unsigned long long)f[fstride + c];
    }
    const unsigned long long k = h1 * 0x1E3779B97F4A7C15ull + h2 + (unsigned long long)f[2 * fstride + prev[i]] + (unsigned long long)xcls[i] * 0x632BE59BD9B4E019ull +
                                 (unsigned long long)(e1 - e0) * 0x2545F4914F6CDD1Dull;
    key[i] = (int64_t)(k >> (64 - key_bits));          // the top key_bits (<= 63: non-negative, torch sorts signed values)
}

__global__ __launch_bounds__(256) void k_colour_check(int64_t N, const int32_t* ptr, const int32_t* idx, const int32_t* prev, const uint8_t* xcls,
                                                      const int32_t* cid, const int32_t* rep, int32_t* flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int64_t r = rep[cid[i]];
    if (r == i) return;
    const int a0 = ptr[i], a1 = ptr[i + 1], b0 = ptr[r], b1 = ptr[r + 1];
    if (xcls[i] != xcls[r] || prev[i] != prev[r] || a1 - a0 != b1 - b0) { flags[0] = 1; return; }
    const int d = a1 - a0;
    if (d > kColourCheckMax) { atomicAdd(flags + 1, 1); return; }
    for (int k = 0; k < d; ++k) {
        const int c = prev[idx[a0 + k]];
        int na = 0, nb = 0;
        for (int m = 0; m < d; ++m) { na += prev[idx[a0 + m]] == c; nb += prev[idx[b0 + m]] == c; }
        if (na != nb) { flags[0] = 1; return; }
    }
}

// ---- a refinement stage's tables from the sorted keys, on the device (GraphPlan._quotient_dev).
// runs of equal keys -> colours: first[i] marks a run's first sorted position, its exclusive scan numbers the runs
__global__ __launch_bounds__(256) void k_run_first(int64_t N, const int64_t* skey, int32_t* first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) first[i] = (i == 0 || skey[i] != skey[i - 1]) ? 1 : 0;
}
// pos = exclusive scan of first (pos[N] = number of runs): colour of sorted position i = pos[i] + first[i] - 1
__global__ __launch_bounds__(256) void k_run_assign(int64_t N, const int32_t* first, const int32_t* pos, const int32_t* by_colour, int32_t* cid,
                                                    int32_t* starts, int32_t* rep, int32_t* n_colours) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int f = first[i], c = pos[i] + f - 1, v = by_colour[i];
    cid[v] = c;
    if (f) { starts[c] = (int32_t)i; rep[c] = v; }
    if (i == N - 1) { starts[c + 1] = (int32_t)N; n_colours[0] = c + 1; }
}
// per colour: its representative's degree, previous colour and feature class; heavy rows counted
__global__ __launch_bounds__(256) void k_rep_rows(int64_t C, const int32_t* rep, const int32_t* ptr, const int32_t* prev, const uint8_t* xcls, int heavy_row,
                                                  int32_t* dr, int32_t* own, uint8_t* xrep, int32_t* n_heavy) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r = rep[c], d = ptr[r + 1] - ptr[r];
    dr[c] = d; own[c] = prev[r]; xrep[c] = xcls[r];
    if (d > heavy_row) atomicAdd(n_heavy, 1);
}
// a representative's list in previous colours, and the colour that owns every list entry
__global__ __launch_bounds__(256) void k_rep_lists(int64_t C, const int32_t* rep, const int32_t* ptr, const int32_t* idx, const int32_t* prev,
                                                   const int32_t* rptr, int32_t* ent, int32_t* row) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r = rep[c], e0 = ptr[r], d = ptr[r + 1] - e0, o = rptr[c];
    for (int k = 0; k < d; ++k) { ent[o + k] = prev[idx[e0 + k]]; row[o + k] = (int32_t)c; }
}
// members per key of an ascending key array: counts[k] = lower_bound(k + 1) - lower_bound(k)
__global__ __launch_bounds__(256) void k_sorted_key_counts(int64_t n, const int32_t* sorted, int64_t K, int32_t* counts) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (sorted[m] < k) lo = m + 1; else hi = m; }
    const int64_t a = lo;
    hi = n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (sorted[m] <= k) lo = m + 1; else hi = m; }
    counts[k] = (int32_t)(lo - a);
}
// segment tables of mgv_seg_sum, one level: colour g with counts[g] members is cut into nseg = max(1, ceil(counts / seg)) segments;
// a colour with more than one leaves partial rows (summed by the next level)
__global__ __launch_bounds__(256) void k_seg_level_counts(int64_t G, const int32_t* counts, int seg, int32_t* nseg, int32_t* part, int32_t* multi) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const int n = counts[g], s = n > seg ? (n + seg - 1) / seg : 1;
    nseg[g] = s; part[g] = s > 1 ? s : 0; multi[g] = s > 1;
}
// the same five steps (counts -> nseg / part / multi, their four exclusive scans, the totals) by ONE workgroup: most levels have a few
// thousand colours at most, and 14 launches for them cost more than the work
__global__ __launch_bounds__(1024) void k_seg_level_small(int64_t G, const int32_t* counts, int seg, int32_t* nseg, int32_t* first, int32_t* start,
                                                          int32_t* pfirst, int32_t* midx, int32_t* totals) {
    __shared__ int s_w[16][4];
    __shared__ int s_carry[4];
    if (threadIdx.x < 4) s_carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t g0 = 0; g0 < G; g0 += 1024) {
        const int64_t g = g0 + threadIdx.x;
        int v[4] = {0, 0, 0, 0};
        if (g < G) {
            const int n = counts[g], s = n > seg ? (n + seg - 1) / seg : 1;
            nseg[g] = s;
            v[0] = s; v[1] = n; v[2] = s > 1 ? s : 0; v[3] = s > 1;
        }
        int inc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            inc[c] = v[c];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc[c], d, 64); if (lane >= d) inc[c] += u; }
            if (lane == 63) s_w[w][c] = inc[c];
        }
        __syncthreads();
        int off[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            off[c] = s_carry[c];
            for (int k = 0; k < w; ++k) off[c] += s_w[k][c];
        }
        if (g < G) { first[g] = off[0] + inc[0] - v[0]; start[g] = off[1] + inc[1] - v[1]; pfirst[g] = off[2] + inc[2] - v[2]; midx[g] = off[3] + inc[3] - v[3]; }
        __syncthreads();
        if (threadIdx.x == 1023) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s_carry[c] = off[c] + inc[c];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        first[G] = s_carry[0]; start[G] = s_carry[1]; pfirst[G] = s_carry[2]; midx[G] = s_carry[3];
        totals[0] = s_carry[0]; totals[1] = s_carry[1]; totals[2] = s_carry[2]; totals[3] = s_carry[3];
    }
}
__global__ __launch_bounds__(64) void k_seg_level_totals(int64_t G, const int32_t* first, const int32_t* start, const int32_t* pfirst, const int32_t* midx,
                                                         int32_t* totals) {
    if (threadIdx.x == 0) { totals[0] = first[G]; totals[1] = start[G]; totals[2] = pfirst[G]; totals[3] = midx[G]; }
}
__global__ __launch_bounds__(256) void k_seg_level_fill(int64_t G, int64_t n_seg, const int32_t* gid, int seg, int base, const int32_t* nseg,
                                                        const int32_t* first, const int32_t* start, const int32_t* pfirst, const int32_t* midx,
                                                        int32_t* seg_ptr, int32_t* out_row, int32_t* gid_next, int32_t* counts_next) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n_seg) {
        int64_t lo = 0, hi = G;                     // the colour g with first[g] <= t < first[g + 1] (every colour has >= 1 segment)
        while (hi - lo > 1) { const int64_t m = (lo + hi) >> 1; if (first[m] <= t) lo = m; else hi = m; }
        const int g = (int)lo, j = (int)(t - first[g]);
        seg_ptr[t] = start[g] + seg * j;
        if (out_row) out_row[t] = nseg[g] > 1 ? base + pfirst[g] + j : (gid ? gid[g] : g);
        if (t == 0) seg_ptr[n_seg] = start[G];
    }
    if (t < G && nseg[t] > 1 && gid_next) { gid_next[midx[t]] = gid ? gid[t] : (int32_t)t; counts_next[midx[t]] = nseg[t]; }
}

}  // namespace mgv

using namespace mgv;

extern "C" int mgv_scan_exclusive_i32(int64_t n, const int32_t* in, int32_t* out, int32_t* scratch, void* stream) {
    MGV_CHECK_ARG(n >= 0 && out && scratch && (n == 0 || in));
    return scan_exclusive(n, in, out, scratch, static_cast<hipStream_t>(stream));
}

extern "C" int mgv_plan_csr_scratch_ints(int64_t N, int64_t E) { return (int)(2 * N + 4 * E + (N + E) / kScanItems + 64); }

/* in_eid / out_eid (optional, E ints each): the original edge id behind every in- / out-CSR slot.
 * scratch (int32): cur_in[N] cur_out[N] eid_in[E] eid_out[E] pos_in[E] tmp[E] scan[...] flags[8]; flags are returned in status[0..1]
 * (status[0]: 1 = node id out of range; device memory, 2 ints, written asynchronously) */
extern "C" int mgv_plan_csr(int64_t N, int64_t E, const int64_t* src, const int64_t* dst, int32_t* in_ptr, int32_t* in_src, int32_t* in_dst,
                            int32_t* out_ptr, int32_t* out_dst, int32_t* out_slot, int32_t* in_eid, int32_t* out_eid, int32_t* scratch,
                            int64_t scratch_ints, int32_t* status, void* stream) {
    MGV_CHECK_ARG(N >= 0 && E >= 0 && N < (1LL << 31) && E < (1LL << 31) && in_ptr && out_ptr && scratch && status);
    MGV_CHECK_ARG(scratch_ints >= mgv_plan_csr_scratch_ints(N, E));
    MGV_CHECK_ARG(E == 0 || (src && dst && in_src && in_dst && out_dst && out_slot));
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t* cur_in = scratch;
    int32_t* cur_out = cur_in + N;
    int32_t* eid_in = cur_out + N;
    int32_t* eid_out = eid_in + E;
    int32_t* pos_in = eid_out + E;
    int32_t* tmp = pos_in + E;
    int32_t* scan = tmp + E;
    hipMemsetAsync(cur_in, 0, 2 * N * sizeof(int32_t), st);
    hipMemsetAsync(status, 0, 2 * sizeof(int32_t), st);
    if (E > 0) hipLaunchKernelGGL(k_csr_count, dim3(blocks_for(E)), dim3(256), 0, st, E, N, src, dst, cur_in, cur_out, status);
    int rc = scan_exclusive(N, cur_in, in_ptr, scan, st);
    if (rc != MGV_OK) return rc;
    rc = scan_exclusive(N, cur_out, out_ptr, scan, st);
    if (rc != MGV_OK) return rc;
    if (E == 0) return MGV_OK;
    hipMemcpyAsync(cur_in, in_ptr, N * sizeof(int32_t), hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(cur_out, out_ptr, N * sizeof(int32_t), hipMemcpyDeviceToDevice, st);
    hipMemsetAsync(eid_in, 0, 2 * E * sizeof(int32_t), st);      // eid_in, eid_out: slots no valid edge fills stay a valid edge id (the copies below)
    hipLaunchKernelGGL(k_csr_fill, dim3(blocks_for(E)), dim3(256), 0, st, E, N, src, dst, cur_in, cur_out, eid_in, eid_out);
    // per-node sorts; the heavy list reuses the cursor arrays (dead after the fill): list at cur_in[0..N), counter in status[1]
    for (int dir = 0; dir < 2; ++dir) {
        int32_t* eid = dir ? eid_out : eid_in;
        const int32_t* ptr = dir ? out_ptr : in_ptr;
        int32_t* heavy = dir ? cur_out : cur_in;
        int32_t* n_heavy = scan + dir;                      // two counters in the (now idle) scan scratch
        hipMemsetAsync(n_heavy, 0, sizeof(int32_t), st);
        hipLaunchKernelGGL(k_sort_lists, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, ptr, eid, heavy, n_heavy);
        hipLaunchKernelGGL(k_sort_heavy, dim3(256), dim3(1024), 0, st, heavy, n_heavy, ptr, eid, tmp);
    }
    if (in_eid) hipMemcpyAsync(in_eid, eid_in, E * sizeof(int32_t), hipMemcpyDeviceToDevice, st);        // original edge id of every CSR slot
    if (out_eid) hipMemcpyAsync(out_eid, eid_out, E * sizeof(int32_t), hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(k_csr_finish_in, dim3(blocks_for(E)), dim3(256), 0, st, E, N, in_ptr, src, dst, eid_in, in_src, in_dst, pos_in);
    hipLaunchKernelGGL(k_csr_finish_out, dim3(blocks_for(E)), dim3(256), 0, st, E, N, out_ptr, dst, eid_out, pos_in, out_dst, out_slot);
    MGV_LAUNCH_RET();
}

/* every list vals[ptr[n] .. ptr[n+1]) ascending, in place (equal values are interchangeable).  Turns lists filled through atomic
 * cursors (mgv_neg_bucket) into lists whose order no longer depends on which thread arrived first.  scratch: N + 1 + E ints. */
extern "C" int mgv_sort_lists_i32(int64_t N, int64_t E, const int32_t* ptr, int32_t* vals, int32_t* scratch, int64_t scratch_ints,
                                  void* stream) {
    MGV_CHECK_ARG(N >= 0 && E >= 0 && (N == 0 || (ptr && scratch && scratch_ints >= N + 1 + E)) && (E == 0 || vals));
    if (N == 0 || E == 0) return MGV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t* heavy = scratch;
    int32_t* n_heavy = scratch + N;
    int32_t* tmp = scratch + N + 1;
    hipMemsetAsync(n_heavy, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(k_sort_lists, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, ptr, vals, heavy, n_heavy);
    hipLaunchKernelGGL(k_sort_heavy, dim3(256), dim3(1024), 0, st, heavy, n_heavy, ptr, vals, tmp);
    MGV_LAUNCH_RET();
}

/* ASAP levels of a DAG from its CSRs (utils/dag_utils.py:10-37 semantics: the round in which all parents have been evaluated).
 * `rounds` launches are enqueued without a host round trip; done[0] accumulates the number of nodes levelised: the caller
 * checks done[0] == N afterwards (fewer: more rounds needed, or a cycle).  scratch: pending[N], frontier[2][N], counts[rounds + 2]. */
extern "C" int mgv_plan_levels(int64_t N, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_dst, int32_t* level, int rounds,
                               int32_t* scratch, int64_t scratch_ints, int32_t* done, void* stream) {
    MGV_CHECK_ARG(N >= 0 && rounds >= 1 && in_ptr && out_ptr && level && scratch && done && scratch_ints >= 3 * N + rounds + 2);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (N == 0) return MGV_OK;
    int32_t* pending = scratch;
    int32_t* fr[2] = {pending + N, pending + 2 * N};
    int32_t* counts = pending + 3 * N;                      // one counter per round (round r fills counts[r + 1])
    hipMemsetAsync(counts, 0, (rounds + 2) * sizeof(int32_t), st);
    hipMemsetAsync(done, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(k_level_seed, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, in_ptr, pending, level, fr[0], counts);
    for (int r = 0; r < rounds; ++r)
        hipLaunchKernelGGL(k_level_step, dim3(512), dim3(256), 0, st, fr[r & 1], counts + r, fr[(r + 1) & 1], counts + r + 1, out_ptr, out_dst, pending,
                           level, r, done);
    MGV_LAUNCH_RET();
}

/* gate ids -> aggregator slots, int32 levels, sort keys; maxlevel[0] receives the largest level (device int, zeroed here) */
extern "C" int mgv_plan_keys(int64_t N, int T, const float* gate, const int64_t* level64, const uint8_t* slot_of_gate_host256, uint8_t* gslot,
                             int32_t* level32, int32_t* key, int32_t* maxlevel, void* stream) {
    MGV_CHECK_ARG(N >= 0 && T >= 1 && T <= 255 && slot_of_gate_host256 && maxlevel && (N == 0 || (gate && level64 && gslot && level32 && key)));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipMemsetAsync(maxlevel, 0, sizeof(int32_t), st);
    if (N == 0) return MGV_OK;
    SlotTable tab;
    for (int i = 0; i < 256; ++i) tab.slot[i] = slot_of_gate_host256[i];
    hipLaunchKernelGGL(k_bucket_keys, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, T, gate, level64, tab, gslot, level32, key, maxlevel);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_plan_check_levels(int64_t E, const int32_t* in_src, const int32_t* in_dst, const uint8_t* gslot, const int32_t* level, int32_t* err,
                                     void* stream) {
    MGV_CHECK_ARG(E >= 0 && err && (E == 0 || (in_src && in_dst && gslot && level)));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipMemsetAsync(err, 0, sizeof(int32_t), st);
    if (E > 0) hipLaunchKernelGGL(k_check_levels, dim3(blocks_for(E)), dim3(256), 0, st, E, in_src, in_dst, gslot, level, err);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_count_sort_scratch_ints(int64_t n, int K) {
    const int64_t B = (n + kSortBlock - 1) / kSortBlock;
    const int64_t m = (int64_t)K * (B < 1 ? 1 : B) + 1;
    return (int)(2 * m + m / kScanItems + 64);
}
/* Stable counting sort of the items 0..n-1 by key[i] in [0, K) (key < 0: item dropped): order[] = item ids by (key, id),
 * key_start[K + 1] = first position of each key.  K <= 8192. */
extern "C" int mgv_count_sort_i32(int64_t n, const int32_t* key, int K, int32_t* order, int32_t* key_start, int32_t* scratch, int64_t scratch_ints,
                                  void* stream) {
    MGV_CHECK_ARG(n >= 0 && K >= 1 && K <= kMaxKeys && key_start && scratch && scratch_ints >= mgv_count_sort_scratch_ints(n, K));
    MGV_CHECK_ARG(n == 0 || (key && order));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int B = (int)((n + kSortBlock - 1) / kSortBlock) < 1 ? 1 : (int)((n + kSortBlock - 1) / kSortBlock);
    const int64_t m = (int64_t)K * B;
    int32_t* cnt = scratch;
    int32_t* base = cnt + m + 1;
    int32_t* scan = base + m + 1;
    hipMemsetAsync(cnt, 0, m * sizeof(int32_t), st);
    if (n > 0) hipLaunchKernelGGL(k_sort_hist, dim3(B), dim3(kSortBlock), 0, st, n, key, K, B, cnt);
    int rc = scan_exclusive(m, cnt, base, scan, st);
    if (rc != MGV_OK) return rc;
    if (n > 0) hipLaunchKernelGGL(k_sort_place, dim3(B), dim3(kSortBlock), 0, st, n, key, K, B, base, order);
    hipLaunchKernelGGL(k_key_starts, dim3((K + 256) / 256), dim3(256), 0, st, K, B, base, key_start);
    MGV_LAUNCH_RET();
}

/* tiles of <= 64 consecutive entries of one key each, from key_start[K + 1] (K = levels * T): ntile/tile_first are scratch of
 * K + 1 ints (+ scan scratch); tile_first[K] = number of tiles (device) */
extern "C" int mgv_plan_tile_counts(int K, const int32_t* key_start, int32_t* ntile, int32_t* tile_first, int32_t* scan_scratch, void* stream) {
    MGV_CHECK_ARG(K >= 1 && key_start && ntile && tile_first && scan_scratch);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_key_counts, dim3((K + 255) / 256), dim3(256), 0, st, K, 1, key_start, ntile);
    return scan_exclusive(K, ntile, tile_first, scan_scratch, st);
}
extern "C" int mgv_plan_tiles(int K, int T, int L, int64_t n_active, const int32_t* key_start, const int32_t* tile_first, const int32_t* order,
                              const int32_t* in_ptr, const int32_t* out_ptr, int32_t* tile_start, int32_t* tile_count, int32_t* tile_slot,
                              int32_t* order_span, int32_t* level_tile_ptr, void* stream) {
    MGV_CHECK_ARG(K >= 1 && T >= 1 && L >= 1 && K == L * T && key_start && tile_first && level_tile_ptr);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_tiles_fill, dim3((K + 255) / 256), dim3(256), 0, st, K, 1, T, key_start, tile_first, tile_start, tile_count, tile_slot);
    if (n_active > 0) hipLaunchKernelGGL(k_order_span, dim3((unsigned)((n_active + 255) / 256)), dim3(256), 0, st, n_active, order, in_ptr, out_ptr,
                                         reinterpret_cast<int4*>(order_span));
    hipLaunchKernelGGL(k_level_tile_ptr, dim3((L + 256) / 256), dim3(256), 0, st, L, T, tile_first, level_tile_ptr);
    MGV_LAUNCH_RET();
}

/* Packed sweep rows (func_level_x3_common.h: kRowInts = 32 ints = one 128-byte line per updated node, in sweep order):
 * [0..3] the CSR spans, [4..7] the first 4 in-edge sources (-1: none), [8..23] the first 8 consumers as (node, in-CSR slot) pairs (node -1: none),
 * [24..25] their gate slots (bytes), [26..31] unused.  Eight threads per row, one 16-byte piece each. */
__global__ void __launch_bounds__(256) k_order_rows(int64_t n_active, const int32_t* __restrict__ order, const int32_t* __restrict__ in_ptr,
                                                    const int32_t* __restrict__ in_src, const int32_t* __restrict__ out_ptr,
                                                    const int32_t* __restrict__ out_dst, const int32_t* __restrict__ out_slot,
                                                    const uint8_t* __restrict__ gslot, int4* __restrict__ rows) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t >> 3;
    const int k = (int)(t & 7);
    if (i >= n_active) return;
    const int v = order[i];
    int4 o = make_int4(-1, -1, -1, -1);
    if (k == 0) {
        o = make_int4(in_ptr[v], in_ptr[v + 1], out_ptr[v], out_ptr[v + 1]);
    } else if (k == 1) {
        const int a = in_ptr[v], b = in_ptr[v + 1];
        if (a + 0 < b) o.x = in_src[a + 0];
        if (a + 1 < b) o.y = in_src[a + 1];
        if (a + 2 < b) o.z = in_src[a + 2];
        if (a + 3 < b) o.w = in_src[a + 3];
    } else if (k < 6) {
        const int a = out_ptr[v] + 2 * (k - 2), b = out_ptr[v + 1];
        if (a < b) { o.x = out_dst[a]; o.y = out_slot[a]; }
        if (a + 1 < b) { o.z = out_dst[a + 1]; o.w = out_slot[a + 1]; }
    } else if (k == 6) {
        const int a = out_ptr[v], b = out_ptr[v + 1];
        uint32_t lo = 0xffffffffu, hi = 0xffffffffu;
        for (int j = 0; j < 8 && a + j < b; ++j) {
            const uint32_t gc = gslot[out_dst[a + j]];
            if (j < 4) lo = (lo & ~(0xffu << (8 * j))) | (gc << (8 * j));
            else hi = (hi & ~(0xffu << (8 * (j - 4)))) | (gc << (8 * (j - 4)));
        }
        o.x = (int)lo; o.y = (int)hi;
    }
    rows[i * 8 + k] = o;
}

extern "C" int mgv_plan_order_rows(int64_t n_active, const int32_t* order, const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                   const int32_t* out_dst, const int32_t* out_slot, const uint8_t* gslot, int32_t* rows, void* stream) {
    MGV_CHECK_ARG(n_active >= 0);
    if (n_active == 0) return MGV_OK;
    MGV_CHECK_ARG(order && in_ptr && out_ptr && gslot && rows && (reinterpret_cast<uintptr_t>(rows) & 15) == 0);
    hipLaunchKernelGGL(k_order_rows, dim3((unsigned)((n_active * 8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n_active, order,
                       in_ptr, in_src, out_ptr, out_dst, out_slot, gslot, reinterpret_cast<int4*>(rows));
    MGV_LAUNCH_RET();
}

/* (in-degree, feature class) pairs -> class ids: present/rank are scratch of 65,537 ints each (+ scan scratch of 64); n_cls = rank[65536]
 * (device); status[0] = 3 when a degree exceeds 255 (caller falls back to per-node launches) */
extern "C" int mgv_plan_pairs(int64_t N, const int32_t* in_ptr, const uint8_t* xcls, int32_t* present, int32_t* rank, int32_t* scan_scratch, int32_t* cid,
                              int32_t* cls_deg, uint8_t* cls_x, int32_t* status, void* stream) {
    MGV_CHECK_ARG(N >= 1 && in_ptr && xcls && present && rank && scan_scratch && cid && cls_deg && cls_x && status);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipMemsetAsync(present, 0, 65536 * sizeof(int32_t), st);
    hipMemsetAsync(status, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(k_pair_mark, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, in_ptr, xcls, present, status);
    const int rc = scan_exclusive(65536, present, rank, scan_scratch, st);
    if (rc != MGV_OK) return rc;
    hipLaunchKernelGGL(k_pair_ids, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, in_ptr, xcls, rank, cid);
    hipLaunchKernelGGL(k_pair_table, dim3(256), dim3(256), 0, st, present, rank, cls_deg, cls_x);
    MGV_LAUNCH_RET();
}

/* Colour refinement of GraphPlan.quotient (ops.StructEncoderFn's quotient stages).  keys: key[i] = 63-bit grouping key of node i from
 * its feature class, previous colour prev[i], degree and the multiset of its neighbours' previous colours (f = int64 [3][fstride]
 * random values per previous colour), cut to its top key_bits bits (the radix sort behind it then runs over key_bits: GraphPlan._key_bits
 * sizes them so that two of the expected colours collide with probability < 2^-20; a collision costs speed, never correctness).  check: flags[0] = 1 if some node differs from its group's representative rep[cid[i]] in class,
 * previous colour, degree or neighbour-colour multiset (exact comparison); flags[1] = number of nodes whose lists are longer than 48
 * entries and were NOT compared (the caller checks those by sorting).  flags must be zeroed by the caller. */
extern "C" int mgv_colour_keys(int64_t N, const int32_t* nbr_ptr, const int32_t* nbr_idx, const int32_t* prev, const int64_t* f, int64_t fstride,
                               const uint8_t* xcls, int key_bits, int64_t* key, void* stream) {
    MGV_CHECK_ARG(N >= 0 && nbr_ptr && prev && f && xcls && key && key_bits >= 16 && key_bits <= 63);
    if (N == 0) return MGV_OK;
    hipLaunchKernelGGL(k_colour_keys, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), N, nbr_ptr, nbr_idx, prev, f, fstride, xcls, key_bits, key);
    MGV_LAUNCH_RET();
}
extern "C" int mgv_colour_check(int64_t N, const int32_t* nbr_ptr, const int32_t* nbr_idx, const int32_t* prev, const uint8_t* xcls, const int32_t* cid,
                                const int32_t* rep, int32_t* flags, void* stream) {
    MGV_CHECK_ARG(N >= 0 && nbr_ptr && prev && xcls && cid && rep && flags);
    if (N == 0) return MGV_OK;
    hipLaunchKernelGGL(k_colour_check, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), N, nbr_ptr, nbr_idx, prev, xcls, cid, rep, flags);
    MGV_LAUNCH_RET();
}

/* ---- a colour-refinement stage's tables built on the device (GraphPlan._quotient_dev; the torch composition of the same tables stays in
 * GraphPlan.quotient for CPU plans, and the GPU tests compare the two field by field).
 * groups: from the sorted keys skey[N] and the sorting permutation by_colour[N]: cid[v] = colour (rank of its key's run), starts[C + 1] = first
 * sorted position of every run (starts[C] = N), rep[c] = first member, n_colours[0] = C.  starts / rep hold N + 1 / N ints (C is not known to
 * the host yet); scratch: 2 N + N / 2048 + 66 ints. */
extern "C" int mgv_colour_groups(int64_t N, const int64_t* skey, const int32_t* by_colour, int32_t* cid, int32_t* starts, int32_t* rep, int32_t* n_colours,
                                 int32_t* scratch, int64_t scratch_ints, void* stream) {
    MGV_CHECK_ARG(N >= 1 && skey && by_colour && cid && starts && rep && n_colours && scratch && scratch_ints >= 2 * N + N / kScanItems + 66);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t* first = scratch;
    int32_t* pos = scratch + N;
    int32_t* scan = pos + N + 1;
    const unsigned nb = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(k_run_first, dim3(nb), dim3(256), 0, st, N, skey, first);
    const int rc = scan_exclusive(N, first, pos, scan, st);
    if (rc != MGV_OK) return rc;
    hipLaunchKernelGGL(k_run_assign, dim3(nb), dim3(256), 0, st, N, first, pos, by_colour, cid, starts, rep, n_colours);
    MGV_LAUNCH_RET();
}
/* rep_rows: per colour its representative's previous colour own[C], feature class xrep[C] and list offsets rptr[C + 1] (rptr[C] = number of
 * list entries), n_heavy[0] = colours whose list is longer than heavy_row.  scratch: C + C / 2048 + 66 ints.
 * rep_lists: ent[rptr[c] + k] = previous colour of the representative's k-th neighbour, row[...] = c. */
extern "C" int mgv_colour_rep_rows(int64_t C, const int32_t* rep, const int32_t* nbr_ptr, const int32_t* prev, const uint8_t* xcls, int heavy_row,
                                   int32_t* rptr, int32_t* own, uint8_t* xrep, int32_t* n_heavy, int32_t* scratch, int64_t scratch_ints, void* stream) {
    MGV_CHECK_ARG(C >= 1 && rep && nbr_ptr && prev && xcls && rptr && own && xrep && n_heavy && scratch && scratch_ints >= C + C / kScanItems + 66);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipMemsetAsync(n_heavy, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(k_rep_rows, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, C, rep, nbr_ptr, prev, xcls, heavy_row, scratch, own, xrep, n_heavy);
    return scan_exclusive(C, scratch, rptr, scratch + C, st);
}
extern "C" int mgv_colour_rep_lists(int64_t C, const int32_t* rep, const int32_t* nbr_ptr, const int32_t* nbr_idx, const int32_t* prev, const int32_t* rptr,
                                    int32_t* ent, int32_t* row, void* stream) {
    MGV_CHECK_ARG(C >= 1 && rep && nbr_ptr && prev && rptr && ent && row);
    hipLaunchKernelGGL(k_rep_lists, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), C, rep, nbr_ptr, nbr_idx, prev, rptr, ent, row);
    MGV_LAUNCH_RET();
}
/* members per key k in [0, K) of an ASCENDING key array */
extern "C" int mgv_sorted_key_counts(int64_t n, const int32_t* sorted_keys, int64_t K, int32_t* counts, void* stream) {
    MGV_CHECK_ARG(n >= 0 && K >= 1 && counts && (n == 0 || sorted_keys));
    hipLaunchKernelGGL(k_sorted_key_counts, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n, sorted_keys, K, counts);
    MGV_LAUNCH_RET();
}
/* One level of mgv_seg_sum's segment tables (GraphPlan.class_sum_levels) for G colours with counts[G] members each, cut into segments of <= seg.
 * scan: work[0..3] = {segments, members, partial rows this level leaves, colours with more than one segment}; the rest of `work`
 * (mgv_seg_level_work_ints(G) ints) carries the per-colour scans to `fill`, which writes seg_ptr[n_seg + 1], out_row[n_seg] (NULL: segment s
 * writes row s) — a colour's only segment writes row gid[g] (NULL gid: g), a partial row base + its running number — and the next level's
 * colours gid_next / counts_next (NULL when no colour has more than one segment). */
extern "C" int mgv_seg_level_work_ints(int64_t G) { return (int)(4 + 3 * G + 4 * (G + 1) + G / kScanItems + 66); }
extern "C" int mgv_seg_level_scan(int64_t G, const int32_t* counts, int seg, int32_t* work, int64_t work_ints, void* stream) {
    MGV_CHECK_ARG(G >= 1 && counts && seg >= 2 && work && work_ints >= mgv_seg_level_work_ints(G));
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t* nseg = work + 4; int32_t* part = nseg + G; int32_t* multi = part + G;
    int32_t* first = multi + G; int32_t* start = first + G + 1; int32_t* pfirst = start + G + 1; int32_t* midx = pfirst + G + 1;
    int32_t* scan = midx + G + 1;
    if (G <= 65536) {
        hipLaunchKernelGGL(k_seg_level_small, dim3(1), dim3(1024), 0, st, G, counts, seg, nseg, first, start, pfirst, midx, work);
        MGV_LAUNCH_RET();
    }
    hipLaunchKernelGGL(k_seg_level_counts, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, G, counts, seg, nseg, part, multi);
    int rc = scan_exclusive(G, nseg, first, scan, st);
    if (rc == MGV_OK) rc = scan_exclusive(G, counts, start, scan, st);
    if (rc == MGV_OK) rc = scan_exclusive(G, part, pfirst, scan, st);
    if (rc == MGV_OK) rc = scan_exclusive(G, multi, midx, scan, st);
    if (rc != MGV_OK) return rc;
    hipLaunchKernelGGL(k_seg_level_totals, dim3(1), dim3(64), 0, st, G, first, start, pfirst, midx, work);
    MGV_LAUNCH_RET();
}
extern "C" int mgv_seg_level_fill(int64_t G, int64_t n_seg, const int32_t* gid, int seg, int base, const int32_t* work, int32_t* seg_ptr, int32_t* out_row,
                                  int32_t* gid_next, int32_t* counts_next, void* stream) {
    MGV_CHECK_ARG(G >= 1 && n_seg >= G && work && seg_ptr && ((gid_next == nullptr) == (counts_next == nullptr)));
    const int32_t* nseg = work + 4; const int32_t* first = nseg + 3 * G; const int32_t* start = first + G + 1; const int32_t* pfirst = start + G + 1;
    const int32_t* midx = pfirst + G + 1;
    const int64_t threads = n_seg > G ? n_seg : G;
    hipLaunchKernelGGL(k_seg_level_fill, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), G, n_seg, gid, seg, base, nseg,
                       first, start, pfirst, midx, seg_ptr, out_row, gid_next, counts_next);
    MGV_LAUNCH_RET();
}
