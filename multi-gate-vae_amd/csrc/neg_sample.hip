// Negative edges of the reconstruction loss, sampled and bucketed on the device.
//
// The reference calls torch_geometric.utils.negative_sampling(pos + self loops, N) (dg_ae_model_aig.py:115-119):
// as many (src, dst) pairs as that edge set has, uniform over the pairs that are neither an edge nor a self loop.
// Here every output slot draws pairs from a counter-based generator until one passes the test (the edge test walks
// the source's out-list in the batch's CSR: no sorted key table, no compaction pass), and counts its two ends so
// that a second pass can bucket the pairs by source and by destination: the backward of the loss then gathers
// instead of adding 2 x 256 bytes per pair with memory-side float atomics.
#include "mgv_common.h"
#include "../../include/mgvae_hip.h"

namespace mgv {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(kThreads) void k_neg_sample(int64_t N, int64_t E, uint64_t seed, const int32_t* out_ptr, const int32_t* out_dst,
                                                        int64_t* neg_src, int64_t* neg_dst, int32_t* cnt_out, int32_t* cnt_in,
                                                        int32_t* rank_out, int32_t* rank_in) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < E; i += (int64_t)gridDim.x * kThreads) {
        uint64_t ctr = seed ^ ((uint64_t)i * 0xD1342543DE82EF95ull);
        int s, d;
        for (int attempt = 0;; ++attempt) {
            const uint64_t r = splitmix64(ctr + attempt);
            s = (int)(((r >> 32) * (uint64_t)N) >> 32);
            d = (int)(((r & 0xFFFFFFFFull) * (uint64_t)N) >> 32);
            bool ok = s != d;
            if (ok)
                for (int e = out_ptr[s]; e < out_ptr[s + 1]; ++e)
                    if (out_dst[e] == d) { ok = false; break; }
            if (ok || attempt >= 64) break;      // 64 straight rejections: the graph is (nearly) complete; keep the pair
        }
        neg_src[i] = s; neg_dst[i] = d;
        // the pair's place inside its source's and its destination's bucket: the count it found (the bucket pass then needs no atomics)
        rank_out[i] = atomicAdd(cnt_out + s, 1);
        rank_in[i] = atomicAdd(cnt_in + d, 1);
    }
}

// bucket pass: ptr arrays = exclusive scans of the counts, rank_* = the pairs' places inside their buckets (from the sampling pass)
__global__ __launch_bounds__(kThreads) void k_neg_bucket(int64_t E, const int64_t* neg_src, const int64_t* neg_dst, const int32_t* out_ptr,
                                                        const int32_t* in_ptr, const int32_t* rank_out, const int32_t* rank_in, int64_t* srt_src,
                                                        int64_t* srt_dst, int32_t* out_dst, int32_t* in_src) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < E; i += (int64_t)gridDim.x * kThreads) {
        const int s = (int)neg_src[i], d = (int)neg_dst[i];
        const int p = out_ptr[s] + rank_out[i];
        srt_src[p] = s; srt_dst[p] = d; out_dst[p] = d;
        in_src[in_ptr[d] + rank_in[i]] = s;
    }
}

}  // namespace mgv

extern "C" int mgv_neg_sample(int64_t N, int64_t E, uint64_t seed, const int32_t* pos_out_ptr, const int32_t* pos_out_dst,
                              int64_t* neg_src, int64_t* neg_dst, int32_t* cnt_out, int32_t* cnt_in, int32_t* rank_out, int32_t* rank_in,
                              void* stream) {
    MGV_CHECK_ARG(N >= 2 && N < (1ll << 31) && E >= 0 && pos_out_ptr && neg_src && neg_dst && cnt_out && cnt_in && rank_out && rank_in);
    if (E == 0) return MGV_OK;
    hipLaunchKernelGGL(mgv::k_neg_sample, dim3(mgv::grid_for((E + mgv::kThreads - 1) / mgv::kThreads, 16)), dim3(mgv::kThreads), 0,
                       static_cast<hipStream_t>(stream), N, E, seed, pos_out_ptr, pos_out_dst, neg_src, neg_dst, cnt_out, cnt_in, rank_out, rank_in);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_neg_bucket(int64_t E, const int64_t* neg_src, const int64_t* neg_dst, const int32_t* out_ptr, const int32_t* in_ptr,
                              const int32_t* rank_out, const int32_t* rank_in, int64_t* srt_src, int64_t* srt_dst, int32_t* out_dst, int32_t* in_src,
                              void* stream) {
    MGV_CHECK_ARG(E >= 0 && out_ptr && in_ptr && rank_out && rank_in);
    if (E == 0) return MGV_OK;
    MGV_CHECK_ARG(neg_src && neg_dst && srt_src && srt_dst && out_dst && in_src);
    hipLaunchKernelGGL(mgv::k_neg_bucket, dim3(mgv::grid_for((E + mgv::kThreads - 1) / mgv::kThreads, 16)), dim3(mgv::kThreads), 0,
                       static_cast<hipStream_t>(stream), E, neg_src, neg_dst, out_ptr, in_ptr, rank_out, rank_in, srt_src, srt_dst, out_dst, in_src);
    MGV_LAUNCH_RET();
}
