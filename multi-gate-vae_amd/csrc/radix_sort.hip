// Stable radix sort of (key, position) pairs for the plan builder's colour refinement (GraphPlan.quotient): rocPRIM's device radix
// sort with a counting iterator as the value input, so the permutation comes out as int32 without an index array being written
// and read first (torch.sort moves int64 indices and sorts all 64 key bits whatever the key range).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include "mgv_common.h"
#include "../../include/mgvae_hip.h"

namespace {
template <typename K>
hipError_t sort_iota(void* temp, size_t& bytes, const K* kin, K* kout, int32_t* order, size_t n, int end_bit, hipStream_t st) {
    rocprim::counting_iterator<int32_t> iota(0);
    return rocprim::radix_sort_pairs(temp, bytes, kin, kout, iota, order, n, 0u, (unsigned)end_bit, st);
}
}  // namespace

extern "C" int mgv_sort_pairs_temp_ints(int key_bytes, int64_t n) {
    size_t bytes = 0;
    hipError_t e = hipErrorInvalidValue;
    if (key_bytes == 8) e = sort_iota<uint64_t>(nullptr, bytes, nullptr, nullptr, nullptr, (size_t)(n < 1 ? 1 : n), 64, nullptr);
    else if (key_bytes == 4) e = sort_iota<uint32_t>(nullptr, bytes, nullptr, nullptr, nullptr, (size_t)(n < 1 ? 1 : n), 32, nullptr);
    return (e == hipSuccess && bytes / 4 + 4 < (size_t(1) << 31)) ? (int)(bytes / 4 + 4) : -1;
}

extern "C" int mgv_sort_pairs(int key_bytes, int64_t n, const void* keys_in, void* keys_out, int32_t* order, int end_bit, void* temp,
                              int64_t temp_ints, void* stream) {
    MGV_CHECK_ARG(n >= 0 && n < (int64_t(1) << 31) && (key_bytes == 4 || key_bytes == 8) && end_bit >= 1 && end_bit <= 8 * key_bytes);
    if (n == 0) return MGV_OK;
    MGV_CHECK_ARG(keys_in && keys_out && order && temp && temp_ints >= mgv_sort_pairs_temp_ints(key_bytes, n));
    size_t bytes = (size_t)temp_ints * 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const hipError_t e = key_bytes == 8
        ? sort_iota<uint64_t>(temp, bytes, static_cast<const uint64_t*>(keys_in), static_cast<uint64_t*>(keys_out), order, (size_t)n, end_bit, st)
        : sort_iota<uint32_t>(temp, bytes, static_cast<const uint32_t*>(keys_in), static_cast<uint32_t*>(keys_out), order, (size_t)n, end_bit, st);
    return e == hipSuccess ? MGV_OK : (int)e;
}
