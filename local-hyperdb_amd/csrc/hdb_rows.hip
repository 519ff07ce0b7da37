// hdb_rows.hip -- matrix lifecycle kernels: compaction of the resident matrix after remove_document.
//
// Reference: HyperDB.remove_document (hyperdb/hyperdb.py:691-766) rebuilds self.vectors on the host with
// np.vstack / a boolean mask (:721-728) and then re-normalises every row again on the next query.  Here the kept rows
// are gathered on the device, out of place, in ONE pass at HBM speed, and the per-row caches (1/||v||, ||v||^2) travel
// with their rows, so nothing is recomputed and nothing crosses PCIe.
//
// Layout: out[j] = V[rows[j]] for j in [0, m); `rows` is ascending, so both the reads and the writes of a
// workgroup's slice are close to sequential.  One wave moves one row per step, 16 bytes per lane (whole 128-B lines
// for the usual d); rows that are not a multiple of 16 bytes fall back to element copies.
// Algorithmic bytes per kept row: 2 * row_bytes + 16 (row read + row written, two cache floats read + written).
#include "hdb_common.h"

typedef unsigned int hdb_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void hdb_gather_rows_kernel(const char* __restrict__ V, const int64_t* __restrict__ rows, int64_t m,
                                                              int row_bytes, char* __restrict__ out,
                                                              const float* __restrict__ inv_in, const float* __restrict__ sq_in,
                                                              float* __restrict__ inv_out, float* __restrict__ sq_out,
                                                              int* __restrict__ nan_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const bool vec = (row_bytes & 15) == 0;
    for (int64_t j = wave; j < m; j += nwaves) {
        const int64_t r = rows[j];
        const char* src = V + r * (int64_t)row_bytes;
        char* dst = out + j * (int64_t)row_bytes;
        if (vec) {
            for (int c = lane * 16; c < row_bytes; c += 64 * 16)
                __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const hdb_u32x4*>(src + c)),
                                            reinterpret_cast<hdb_u32x4*>(dst + c));
        } else {
            for (int c = lane; c < row_bytes; c += 64) dst[c] = src[c];
        }
        if (lane == 0) {
            const float ss = sq_in[r];
            inv_out[j] = inv_in[r]; sq_out[j] = ss;
            if (ss != ss) atomicOr(nan_flag, 1);           // the NaN flag of the compacted matrix (ranking_algorithm.py:150)
            else if (ss - ss != 0.f) atomicOr(nan_flag, 2); // ... and its "infinite sum of squares" flag (hdb_rownorm_kernel)
        }
    }
}

extern "C" int hdb_launch_gather_rows(const void* V, const int64_t* rows, int64_t m, int row_bytes, void* out, const float* inv_in,
                                      const float* sq_in, float* inv_out, float* sq_out, int* nan_flag, void* stream) {
    if (m <= 0) return 0;
    const int blocks = hdb_grid_for(m, 4 * 4, 256 * 8);
    hipLaunchKernelGGL(hdb_gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)V, rows, m, row_bytes,
                       (char*)out, inv_in, sq_in, inv_out, sq_out, nan_flag);
    return (int)hipGetLastError();
}
