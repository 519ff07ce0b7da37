// hdb_sort.hip -- full-sort path for very large k (k > HDB_MAX_K on a matrix larger than the candidate
// list): "top_k > N returns all N rows sorted" of hyperdb/ranking_algorithm.py:195-200 must also hold when
// N is in the millions.  The N scores of one query are turned into orderable 32-bit keys, paired with their
// row numbers and radix-sorted descending (rocPRIM device radix sort -- library sort for a cold path, the hot
// top-k paths are the hand-written kernels in hdb_scan/hdb_mfma/hdb_select).  The sort is stable and rows
// enter in ascending order, so equal scores stay in ascending row order: the build's tie rule.
#include <cstring>
#include "hdb_common.h"
#include "../../include/hyperdb_hip.h"
#include <rocprim/rocprim.hpp>

__global__ __launch_bounds__(256) void hdb_sortkeys_kernel(const float* scores, int64_t n, uint32_t* keys, uint32_t* vals) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        keys[i] = hdb_f2key(scores[i]);
        vals[i] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void hdb_sortemit_kernel(const uint32_t* keys, const uint32_t* vals, int64_t n, int64_t k,
                                                           int64_t row_base, int64_t* idx_out, float* score_out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < k; i += (int64_t)gridDim.x * 256) {
        if (i < n) { idx_out[i] = row_base + vals[i]; score_out[i] = hdb_key2f(keys[i]); }
        else { idx_out[i] = -1; score_out[i] = -INFINITY; }
    }
}

extern "C" int hdb_sort_temp_bytes(int64_t n, size_t* bytes) {
    uint32_t* p = nullptr;
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, b, p, p, p, p, (size_t)n, 0, 32, (hipStream_t)0);
    *bytes = b;
    return (int)e;
}

// scores: n floats of ONE query.  work: 4*n uint32 (keys in/out, vals in/out).  temp: hdb_sort_temp_bytes(n).
extern "C" int hdb_launch_full_sort(const float* scores, int64_t n, int64_t k, int64_t row_base, uint32_t* work, void* temp,
                                    size_t temp_bytes, int64_t* idx_out, float* score_out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    uint32_t* kin = work; uint32_t* kout = work + n; uint32_t* vin = work + 2 * n; uint32_t* vout = work + 3 * n;
    hipLaunchKernelGGL(hdb_sortkeys_kernel, dim3(hdb_grid_for(n, 256, 2048)), dim3(256), 0, st, scores, n, kin, vin);
    hipError_t e = rocprim::radix_sort_pairs_desc(temp, temp_bytes, kin, kout, vin, vout, (size_t)n, 0, 32, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(hdb_sortemit_kernel, dim3(hdb_grid_for(k, 256, 2048)), dim3(256), 0, st, kout, vout, n, k, row_base, idx_out, score_out);
    return (int)hipGetLastError();
}
