// hdb_scan.hip -- HBM-bound row-scan kernels (VALU path) for gfx950.
//
// One pass over the stored N x d matrix computes, for QT queries at a time, the metric score of
// every row (reference: the numpy metric functions of hyperdb/ranking_algorithm.py:24-52 and
// :128-147) and either writes it (MODE 0: per-metric functions, row sample, exact path) or
// compares it with a per-query threshold and appends survivors to a candidate list (MODE 1: the
// fused top-k of ranking_algorithm.py:194-200 -- the N-sized score vector is never written).
//
// Layout in HBM: V is the caller's C-contiguous N x d array (HyperDB.vectors).  A wave covers a
// tile of 16 consecutive rows: lane group g = lane>>4 owns rows 4g..4g+3 of the tile and its 16
// lanes stride over the row in 16-byte chunks (256 contiguous bytes per group per load, whole
// 128-B lines; a d=384 fp16 row is exactly 3 such loads).  Queries sit in LDS in the accumulate
// type; per-row sums are finished with four DPP row rotations (no LDS, no shuffles).
// Algorithmic bytes per row = row_bytes (+4 for the cached 1/||v||, +4 bias when set).
#include "hdb_common.h"
#include "../../include/hyperdb_hip.h"

template <typename T> struct Elem;
template <> struct Elem<__half> { using Acc = float;  static constexpr int EPC = 8; };
template <> struct Elem<float>  { using Acc = float;  static constexpr int EPC = 4; };
template <> struct Elem<double> { using Acc = double; static constexpr int EPC = 2; };

__device__ __forceinline__ void hdb_unpack(const uint4& raw, float (&x)[8], __half*) {
    const __half2* h = reinterpret_cast<const __half2*>(&raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) { float2 f = __half22float2(h[i]); x[2 * i] = f.x; x[2 * i + 1] = f.y; }
}
__device__ __forceinline__ void hdb_unpack(const uint4& raw, float (&x)[4], float*) {
    x[0] = __uint_as_float(raw.x); x[1] = __uint_as_float(raw.y);
    x[2] = __uint_as_float(raw.z); x[3] = __uint_as_float(raw.w);
}
__device__ __forceinline__ void hdb_unpack(const uint4& raw, double (&x)[2], double*) {
    x[0] = __longlong_as_double(((unsigned long long)raw.y << 32) | raw.x);
    x[1] = __longlong_as_double(((unsigned long long)raw.w << 32) | raw.z);
}

__device__ __forceinline__ float hdb_to_f(__half v) { return __half2float(v); }
__device__ __forceinline__ float hdb_to_f(float v) { return v; }
__device__ __forceinline__ double hdb_to_f(double v) { return v; }

// Survivors of a filter pass (MODE 1) wait in LDS, per block and query slot, and reach the global candidate lists with ONE atomic
// per block and query at the end of the block's tiles.  A returning atomic per survivor on the per-query counters (one cache line:
// ~88 per us) was what small matrices spent their time in: n = 200k x 100 float32, 16 queries, ~2 k survivors each -- 374 us of
// filter pass for 11 us of matrix; the N <= 8192 path, where every row survives, likewise (tools/diag_small_valu.py).
#define HDB_STAGE_CAP 128
struct HdbStage {
    unsigned long long buf[4][HDB_STAGE_CAP];      // up to four query slots per block (QT, QH <= 4)
    unsigned int cnt[4];
    unsigned int base;
};
__device__ __forceinline__ void hdb_stage_init(HdbStage& st) { if (threadIdx.x < 4) st.cnt[threadIdx.x] = 0u; }   // (before a block barrier)
// every thread of the block, after its last tile; slot s holds the survivors of local query ql0 + s
__device__ __forceinline__ void hdb_stage_flush(const ScanArgs& a, HdbStage& st, int ql0, int nslots) {
    __syncthreads();
    for (int s = 0; s < nslots; ++s) {
        const unsigned int have = min(st.cnt[s], (unsigned int)HDB_STAGE_CAP);        // (block-uniform)
        if (have == 0u) continue;
        if (threadIdx.x == 0) st.base = atomicAdd(&a.cnt[(ql0 + s) * HDB_CNT_STRIDE], have);
        __syncthreads();
        const unsigned int base = st.base;
        for (unsigned int e = threadIdx.x; e < have; e += blockDim.x)
            if (base + e < a.cap) a.cand[(int64_t)(ql0 + s) * a.cap + base + e] = st.buf[s][e];
        __syncthreads();
    }
}

// Shared epilogue: raw row sum -> final score of hyperDB_ranking_algorithm_sort, then store / filter.
template <int MODE, typename Acc>
__device__ __forceinline__ void hdb_emit(const ScanArgs& a, int q, int64_t row, int64_t out_i, Acc sum, HdbStage* stg = nullptr, int slot = 0) {
    float s;
    if (a.metric == HDB_EUCLIDEAN) {
        s = (float)(Acc(1) / (Acc(1) + sqrt(sum)));                       // 1/(1+||v-q||), :49-51
    } else if (a.metric == HDB_MANHATTAN) {
        s = (float)(Acc(1) / (Acc(1) + sum));                             // 1/(1+sum|v-q|), :59-60
    } else if (a.metric == HDB_EUCLIDEAN_DIST) {
        s = (float)sqrt(sum);
    } else if (a.metric == HDB_COSINE) {
        s = (float)sum * a.inv_norm[row] * a.qinv[q];                      // :37-41 with cached norms
    } else {
        s = (float)sum;                                                    // dot / hamming
    }
    if (a.bias) s += a.bias[row];                                          // recency, :186
    const bool masked = a.mask && !a.mask[row];                            // filtered-out row
    if (masked) s = -INFINITY;
    if (!a.raw) s = hdb_canon(s);                                          // NaN -> -inf, :174
    if (MODE == 0) {
        a.scores[(int64_t)(q - a.q0) * a.ld + out_i] = s;
    } else {
        const int ql = q - a.q0;
        if (!masked && s >= a.thr[ql]) {
            const unsigned long long ent = hdb_pack(s, (uint32_t)row);
            if (stg) {
                const unsigned int lp = atomicAdd(&stg->cnt[slot], 1u);              // LDS
                if (lp < HDB_STAGE_CAP) { stg->buf[slot][lp] = ent; return; }
            }
            const uint32_t pos = atomicAdd(&a.cnt[ql * HDB_CNT_STRIDE], 1u);         // the slot is full: straight to the global list
            if (pos < a.cap) a.cand[(int64_t)ql * a.cap + pos] = ent;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Vector scan: rows are multiples of 16 bytes and V is 16-byte aligned.
// grid = (blocks, ceil(nq_launch / QT)); block = 256 threads = 4 waves; LDS = QT*d*sizeof(Acc).
// NJ > 0: compile-time number of 16-chunk steps per row (d*sizeof(T) == NJ*256), fully unrolled so
// that all 4*NJ loads of a tile are in flight together; NJ == 0: runtime loop, any nchunks.
// ------------------------------------------------------------------------------------------------
template <typename T, int QT, int MODE, int ACC, int NJ>
__global__ __launch_bounds__(256) void hdb_scan_kernel(ScanArgs a, int nq_end) {
    using Acc = typename Elem<T>::Acc;
    constexpr int EPC = Elem<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Acc* qs = reinterpret_cast<Acc*>(smem);
    __shared__ HdbStage stage;
    if (MODE == 1) hdb_stage_init(stage);

    const int qbase = a.q0 + blockIdx.y * QT;
    for (int i = threadIdx.x; i < QT * a.d; i += 256) {
        const int qt = i / a.d, e = i - qt * a.d;
        const int q = min(qbase + qt, nq_end - 1);
        qs[i] = reinterpret_cast<const Acc*>(a.Q)[(int64_t)q * a.d + e];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int nj = NJ > 0 ? NJ : (a.nchunks + 15) >> 4;
    const char* Vb = reinterpret_cast<const char*>(a.V);

    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < a.ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t r0 = hdb_tile_index(t, a.tile_stride) * 16 + 4 * g;
        const char* p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = min(r0 + u, a.n - 1);
            p[u] = Vb + r * (int64_t)a.row_bytes;
        }
        Acc acc[4][QT];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) acc[u][qt] = Acc(0);

        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        // one 16-byte chunk of each of the 4 rows: non-temporal (V is streamed exactly once per query)
        auto load4 = [&](int cc, uint4 (&raw)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p[u] + (int64_t)cc * 16));
                raw[u] = make_uint4(v.x, v.y, v.z, v.w);
            }
        };
        auto fma4 = [&](int cc, bool live, const uint4 (&raw)[4]) {
            Acc x[4][EPC];
#pragma unroll
            for (int u = 0; u < 4; ++u) hdb_unpack(raw[u], x[u], (T*)nullptr);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                Acc qv[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) qv[e] = qs[qt * a.d + cc * EPC + e];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        // lanes past the end of a ragged row (live == false) contribute exactly 0
                        if (ACC == 1) {
                            const Acc df = live ? x[u][e] - qv[e] : Acc(0);
                            acc[u][qt] += df * df;
                        } else if (ACC == 2) {
                            const Acc df = live ? x[u][e] - qv[e] : Acc(0);
                            acc[u][qt] += df < Acc(0) ? -df : df;
                        } else {
                            acc[u][qt] += (live ? x[u][e] : Acc(0)) * qv[e];
                        }
                    }
                }
            }
        };
        if constexpr (NJ > 0) {
            // rows of exactly NJ*256 bytes: issue ALL loads of the tile (4*NJ x 16 B per lane, up to 3 steps at a
            // time) before the first use, so that every wave keeps 12 KiB in flight whatever the compiler schedules
            constexpr int G = (NJ > 3 && QT > 1) ? 3 : NJ;
#pragma unroll
            for (int j0 = 0; j0 < NJ; j0 += G) {
                uint4 raw[G][4];
#pragma unroll
                for (int j = 0; j < G; ++j) load4(l16 + 16 * (j0 + j), raw[j]);
                __builtin_amdgcn_sched_barrier(0);       // keep the loads ahead of every use (hipcc otherwise re-serialises them)
#pragma unroll
                for (int j = 0; j < G; ++j) fma4(l16 + 16 * (j0 + j), true, raw[j]);
            }
        } else {
            for (int j = 0; j < nj; ++j) {
                const int c = l16 + 16 * j;
                const bool live = c < a.nchunks;
                const int cc = live ? c : a.nchunks - 1;
                uint4 raw[4];
                load4(cc, raw);
                fma4(cc, live, raw);
            }
        }
        // finish the 16-lane row sums; lane l16==u keeps row u of its group
        const int u_own = hdb_owned_row(l16);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const Acc mine = hdb_rows4_sum(acc[0][qt], acc[1][qt], acc[2][qt], acc[3][qt], l16);
            const int64_t row = r0 + u_own;
            const int q = qbase + qt;
            if ((l16 & 3) == 0 && row < a.n && q < nq_end)
                hdb_emit<MODE>(a, q, row, t * 16 + 4 * g + u_own, mine, MODE == 1 ? &stage : nullptr, qt);
        }
    }
    if (MODE == 1) hdb_stage_flush(a, stage, (int)blockIdx.y * QT, min(QT, nq_end - qbase));
}

// ------------------------------------------------------------------------------------------------
// Generic scan: any d / alignment (element-wise loads).  Same tiling, one query per launch row.
// ------------------------------------------------------------------------------------------------
template <typename T, int MODE, int ACC>
__global__ __launch_bounds__(256) void hdb_scan_generic_kernel(ScanArgs a, int nq_end) {
    using Acc = typename Elem<T>::Acc;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Acc* qs = reinterpret_cast<Acc*>(smem);
    __shared__ HdbStage stage;
    if (MODE == 1) hdb_stage_init(stage);
    const int q = a.q0 + blockIdx.y;
    for (int i = threadIdx.x; i < a.d; i += 256) qs[i] = reinterpret_cast<const Acc*>(a.Q)[(int64_t)q * a.d + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const T* Vt = reinterpret_cast<const T*>(a.V);
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < a.ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t r0 = hdb_tile_index(t, a.tile_stride) * 16 + 4 * g;
        Acc acc[4] = {Acc(0), Acc(0), Acc(0), Acc(0)};
        for (int e = l16; e < a.d; e += 16) {
            const Acc qv = qs[e];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t r = min(r0 + u, a.n - 1);
                const Acc x = (Acc)hdb_to_f(Vt[r * (int64_t)a.d + e]);
                if (ACC == 1) { const Acc df = x - qv; acc[u] += df * df; }
                else if (ACC == 2) { const Acc df = x - qv; acc[u] += df < Acc(0) ? -df : df; }
                else acc[u] += x * qv;
            }
        }
        const int u_own = hdb_owned_row(l16);
        const Acc mine = hdb_rows4_sum(acc[0], acc[1], acc[2], acc[3], l16);
        const int64_t row = r0 + u_own;
        if ((l16 & 3) == 0 && row < a.n && q < nq_end) hdb_emit<MODE>(a, q, row, t * 16 + 4 * g + u_own, mine, MODE == 1 ? &stage : nullptr, 0);
    }
    if (MODE == 1) hdb_stage_flush(a, stage, (int)blockIdx.y, q < nq_end ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// Row caches: 1/||v|| (norm 0 -> 1, ranking_algorithm.py:11-15), ||v||^2, NaN flag (:150).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void hdb_rownorm_kernel(const T* V, int64_t n, int d, float* inv_norm,
                                                          float* sqnorm, int* nan_flag) {
    using Acc = typename Elem<T>::Acc;
    constexpr int EPC = Elem<T>::EPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int64_t ntiles = (n + 15) / 16;
    const bool vec = ((d * (int)sizeof(T)) % 16 == 0) && ((reinterpret_cast<uintptr_t>(V) & 15) == 0);
    const int nchunks = d * (int)sizeof(T) / 16;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t r0 = t * 16 + 4 * g;
        Acc acc[4] = {Acc(0), Acc(0), Acc(0), Acc(0)};
        if (vec) {
            for (int c = l16; c < nchunks; c += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t r = min(r0 + u, n - 1);
                    const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(V) + r * (int64_t)d * sizeof(T) + (int64_t)c * 16);
                    Acc x[EPC];
                    hdb_unpack(raw, x, (T*)nullptr);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[u] += x[e] * x[e];
                }
            }
        } else {
            for (int e = l16; e < d; e += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t r = min(r0 + u, n - 1);
                    const Acc x = (Acc)hdb_to_f(V[r * (int64_t)d + e]);
                    acc[u] += x * x;
                }
            }
        }
        const Acc mine = hdb_rows4_sum(acc[0], acc[1], acc[2], acc[3], l16);
        const int64_t row = r0 + hdb_owned_row(l16);
        if ((l16 & 3) == 0 && row < n) {
            const float ss = (float)mine;
            if (ss != ss) atomicOr(nan_flag, 1);
            else if (ss - ss != 0.f) atomicOr(nan_flag, 2);      // an infinite sum: infinite or huge elements (no bf16-part arithmetic on this matrix)
            sqnorm[row] = ss;
            inv_norm[row] = (mine == Acc(0)) ? 1.0f : (float)(Acc(1) / sqrt(mine));
        }
    }
}

// Per-query prep: qinv = 1/||q|| (0 -> 1), qsq = ||q||^2, NaN flag, and (q16 != nullptr) the scaled fp16 copy of
// the query that the MFMA scan multiplies with, see hdb_q16_scaled.  One wave per query.
// Round 4, matrices of up to 8192 rows (every row is a candidate): the kernel also does what two more launches did -- thr_init /
// cnt_init != nullptr: threshold -inf and an empty candidate list for the query (hdb_fill_thr_kernel); qbits != nullptr: its sign bits
// (hdb_qsign_kernel).  On the reference's own sizes (151 .. 10 000 documents) a launch is 4-5 us of a 25-us call.
template <typename Acc>
__global__ __launch_bounds__(64) void hdb_qprep_kernel(const Acc* Q, int nq, int d, float* qinv, float* qsq, int* qnan, _Float16* q16,
                                                       float* qscl, float* thr_init, uint32_t* cnt_init, uint32_t* qbits, int W) {
    const int q = blockIdx.x;
    if (q >= nq) return;
    Acc s = Acc(0);
    float amax = 0.f;
    if (qbits) {                                // (same words as hdb_qsign_kernel: bit e of the query = x_e > 0)
        for (int e0 = 0; e0 < d; e0 += 64) {
            const int e = e0 + (int)threadIdx.x;
            const unsigned long long m = __ballot(e < d && Q[(int64_t)q * d + e] > Acc(0));
            if (threadIdx.x == 0) {
                qbits[(int64_t)q * W + (e0 >> 5)] = (uint32_t)m;
                if (e0 + 32 < d) qbits[(int64_t)q * W + (e0 >> 5) + 1] = (uint32_t)(m >> 32);
            }
        }
    }
    if (thr_init && threadIdx.x == 0) { thr_init[q] = -INFINITY; cnt_init[q * HDB_CNT_STRIDE] = 0u; }
    for (int e = threadIdx.x; e < d; e += 64) {
        const Acc x = Q[(int64_t)q * d + e];
        s = fma(x, x, s);                       // explicit fma: the fused kernel's prologue must reproduce this sum bit for bit
        amax = fmaxf(amax, fabsf((float)x));
    }
    if (q16) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        const float scale = hdb_q16_scale(amax);
        for (int e = threadIdx.x; e < d; e += 64) q16[(int64_t)q * d + e] = (_Float16)((float)Q[(int64_t)q * d + e] * scale);
        if (threadIdx.x == 0) qscl[q] = 1.f / scale;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) {
        const float ss = (float)s;
        qsq[q] = ss;
        qinv[q] = (s == Acc(0)) ? 1.0f : (float)(Acc(1) / sqrt(s));
        qnan[q] = (ss != ss) ? 1 : (ss - ss != 0.f) ? 2 : 0;       // 2 = an infinite element: hdb_finalize_kernel's inf_status
    }
}

// ------------------------------------------------------------------------------------------------
// Hamming (ranking_algorithm.py:116-147): binarise by x > 0, similarity = d - popcount(v ^ q).
// Sign bits are packed once per matrix in blocks of 256 rows, word-major inside a block (hdb_bits_word, hdb_common.h): a scan
// thread handling rows 4i..4i+3 reads one uint4 per word, fully coalesced, and a wave's W loads are one contiguous piece.
// Algorithmic bytes per row = 4 * ceil(d/32).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void hdb_signpack_kernel(const T* V, int64_t n, int d, int64_t row0, uint32_t* bits) {      // V: row row0 of the matrix, n rows from there
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
    const int W = (d + 31) >> 5;
    for (int64_t r = wave; r < n; r += nwaves) {               // generic d: a row per wave step, 2-byte .. 8-byte loads (eight rows per step
        for (int e0 = 0; e0 < d; e0 += 64) {                    // with sixteen-lane stores measured no faster: 5.2 vs 4.8 ms at 10M x 384 fp16)
            const int e = e0 + lane;
            const bool pos = (e < d) && (hdb_to_f(V[r * (int64_t)d + e]) > 0);
            const unsigned long long m = __ballot(pos);
            if (lane == 0) {
                bits[hdb_bits_word(row0 + r, e0 >> 5, W)] = (uint32_t)m;
                if (e0 + 32 < d) bits[hdb_bits_word(row0 + r, (e0 >> 5) + 1, W)] = (uint32_t)(m >> 32);
            }
        }
    }
}

// The same for rows that are whole 32-bit words of signs (d % 32 == 0: every MFMA geometry): a wave takes eight rows as one flat run
// of 16-byte chunks (six loads of 1 KiB in flight per lane at d = 384 fp16 instead of 2-byte loads), a lane turns its chunk into
// 16 / sizeof(T) sign bits, 32 / that many neighbouring lanes OR them into a word, the first of them stores it.
template <typename T>
__global__ __launch_bounds__(256) void hdb_signpack_wide_kernel(const T* V, int64_t n, int d, int64_t row0, uint32_t* bits) {
    constexpr int EPL = 16 / (int)sizeof(T), LPW = 32 / EPL, NL = 6, RB = 8;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
    const int W = d >> 5;
    const int cpr = d / EPL;                                    // chunks per row
    for (int64_t r8 = wave * RB; r8 < n; r8 += nwaves * RB) {
        const int rows = (int)(n - r8 < RB ? n - r8 : RB);
        const int nch = rows * cpr;                             // chunks of this run
        const uint4* src = reinterpret_cast<const uint4*>(V + r8 * (int64_t)d);
        for (int j0 = 0; j0 < nch; j0 += NL * 64) {
            uint4 x[NL];
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int j = j0 + i * 64 + lane;
                x[i] = j < nch ? src[j] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int j = j0 + i * 64 + lane;
                const T* el = reinterpret_cast<const T*>(&x[i]);
                uint32_t b = 0u;
#pragma unroll
                for (int t = 0; t < EPL; ++t) b |= (hdb_to_f(el[t]) > 0 ? 1u : 0u) << t;
                uint32_t v = b << (EPL * (lane % LPW));
#pragma unroll
                for (int o = 1; o < LPW; o <<= 1) v |= (uint32_t)__shfl_xor((int)v, o, 64);
                const int f = j * EPL;                          // element inside the run (< 8 d)
                int row = 0;
#pragma unroll
                for (int k = 1; k < RB; ++k) row += f >= k * d ? 1 : 0;
                const int wd = (f - row * d) >> 5;
                if ((lane % LPW) == 0 && j < nch) bits[hdb_bits_word(row0 + r8 + row, wd, W)] = v;
            }
        }
    }
}

template <typename Acc>
__global__ __launch_bounds__(64) void hdb_qsign_kernel(const Acc* Q, int nq, int d, int W, uint32_t* qbits) {
    const int q = blockIdx.x, lane = threadIdx.x;
    for (int e0 = 0; e0 < d; e0 += 64) {
        const int e = e0 + lane;
        const bool pos = (e < d) && (Q[(int64_t)q * d + e] > 0);
        const unsigned long long m = __ballot(pos);
        if (lane == 0) {
            qbits[(int64_t)q * W + (e0 >> 5)] = (uint32_t)m;
            if (e0 + 32 < d) qbits[(int64_t)q * W + (e0 >> 5) + 1] = (uint32_t)(m >> 32);
        }
    }
}

// grid = (blocks, ceil(nq / QH)); each thread owns 4 consecutive rows and scores them against QH queries per
// pass over the bit matrix (a batch re-reads the bits once per QH queries, not once per query).  Work items are the
// 4 row-quads of each 16-row tile, so the same strided tile sample as the float scans is available (a.tile_stride).
// JACCARD: |v & q| / |v | q| on the same sign bits (ranking_algorithm.py:63-75), 0/0 -> NaN like numpy.
template <int MODE, bool JACCARD, int QH>
__global__ __launch_bounds__(256) void hdb_hamming_kernel(ScanArgs a, const uint32_t* bits, int64_t npad, int W,
                                                          const uint32_t* qbits, int nq_end) {
    __shared__ uint32_t qb[QH][512];
    __shared__ HdbStage stage;
    if (MODE == 1) hdb_stage_init(stage);
    const int q0 = a.q0 + blockIdx.y * QH;
    for (int i = threadIdx.x; i < QH * W; i += 256) {
        const int qq = i / W, w = i - qq * W;
        qb[qq][w] = qbits[(int64_t)min(q0 + qq, nq_end - 1) * W + w];
    }
    __syncthreads();
    const int64_t nitems = a.ntiles * 4;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < nitems; j += (int64_t)gridDim.x * 256) {
        const int64_t i = hdb_tile_index(j >> 2, a.tile_stride) * 4 + (j & 3);      // row quad in the matrix
        if (4 * i >= npad) continue;                                                // ragged last tile of a tiny matrix
        uint32_t mism[QH][4], uni[QH][4];
#pragma unroll
        for (int qq = 0; qq < QH; ++qq)
#pragma unroll
            for (int u = 0; u < 4; ++u) { mism[qq][u] = 0; uni[qq][u] = 0; }
        for (int w = 0; w < W; ++w) {
            const uint4 v = hdb_bits_load<false>(bits, i, w, W);
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int qq = 0; qq < QH; ++qq) {
                const uint32_t qw = qb[qq][w];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (JACCARD) { mism[qq][u] += __popc(vv[u] & qw); uni[qq][u] += __popc(vv[u] | qw); }
                    else mism[qq][u] += __popc(vv[u] ^ qw);
                }
            }
        }
#pragma unroll
        for (int qq = 0; qq < QH; ++qq) {
            const int q = q0 + qq;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t row = 4 * i + u;
                if (row < a.n && q < nq_end) {
                    const float sc = JACCARD ? (float)mism[qq][u] / (float)uni[qq][u] : (float)(a.d - (int)mism[qq][u]);
                    hdb_emit<MODE>(a, q, row, 4 * j + u, sc, MODE == 1 ? &stage : nullptr, qq);
                }
            }
        }
    }
    if (MODE == 1) hdb_stage_flush(a, stage, (int)blockIdx.y * QH, min(QH, nq_end - q0));
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
template <typename T, int QT, int MODE, int ACC>
static void launch_vec(const ScanArgs& a, int nq_launch, int blocks, hipStream_t st) {
    using Acc = typename Elem<T>::Acc;
    const dim3 grid(blocks, (nq_launch + QT - 1) / QT);
    const size_t lds = (size_t)QT * a.d * sizeof(Acc);
    const int nq_end = a.q0 + nq_launch;
    if constexpr (QT == 1) {   // fully unrolled variants only for one query: with QT=4 they spill
        const int nj = (a.nchunks % 16 == 0) ? a.nchunks / 16 : 0;
        if (nj == 3) { hipLaunchKernelGGL((hdb_scan_kernel<T, QT, MODE, ACC, 3>), grid, dim3(256), lds, st, a, nq_end); return; }
        if (nj == 6) { hipLaunchKernelGGL((hdb_scan_kernel<T, QT, MODE, ACC, 6>), grid, dim3(256), lds, st, a, nq_end); return; }
    }
    hipLaunchKernelGGL((hdb_scan_kernel<T, QT, MODE, ACC, 0>), grid, dim3(256), lds, st, a, nq_end);
}

template <typename T, int MODE>
static void launch_scan_t(const ScanArgs& a, int nq_launch, int blocks, bool vec, hipStream_t st) {
    using Acc = typename Elem<T>::Acc;
    const int accm = (a.metric == HDB_EUCLIDEAN || a.metric == HDB_EUCLIDEAN_DIST) ? 1 : (a.metric == HDB_MANHATTAN ? 2 : 0);
    const size_t lds4 = (size_t)4 * a.d * sizeof(Acc);
    if (!vec) {
        const dim3 grid(blocks, nq_launch);
        const size_t lds = (size_t)a.d * sizeof(Acc);
        const int nq_end = a.q0 + nq_launch;
        if (accm == 1) hipLaunchKernelGGL((hdb_scan_generic_kernel<T, MODE, 1>), grid, dim3(256), lds, st, a, nq_end);
        else if (accm == 2) hipLaunchKernelGGL((hdb_scan_generic_kernel<T, MODE, 2>), grid, dim3(256), lds, st, a, nq_end);
        else hipLaunchKernelGGL((hdb_scan_generic_kernel<T, MODE, 0>), grid, dim3(256), lds, st, a, nq_end);
        return;
    }
    // two queries already pay for the four-query kernel: one pass over V instead of two (float32 d=768, N=1M: 1 098 -> ~530 us)
    // (fp16 manhattan is VALU-bound in the four-query kernel -- 1 825 us for 5M x 384 against 625 per single-query pass: two
    // queries stay with two passes there)
    const bool qt4 = (nq_launch >= 3 || (nq_launch == 2 && !(accm == 2 && sizeof(T) == 2))) && lds4 <= 60 * 1024;
    if (qt4) {
        if (accm == 1) launch_vec<T, 4, MODE, 1>(a, nq_launch, blocks, st);
        else if (accm == 2) launch_vec<T, 4, MODE, 2>(a, nq_launch, blocks, st);
        else launch_vec<T, 4, MODE, 0>(a, nq_launch, blocks, st);
    } else {
        if (accm == 1) launch_vec<T, 1, MODE, 1>(a, nq_launch, blocks, st);
        else if (accm == 2) launch_vec<T, 1, MODE, 2>(a, nq_launch, blocks, st);
        else launch_vec<T, 1, MODE, 0>(a, nq_launch, blocks, st);
    }
}

// Entry used by hdb_api.hip.  mode: 0 = write scores, 1 = threshold filter.
extern "C" int hdb_launch_scan(const ScanArgs* args, int dtype, int mode, int nq_launch, int max_blocks, void* stream) {
    ScanArgs a = *args;
    hipStream_t st = (hipStream_t)stream;
    const int elem = dtype == HDB_F16 ? 2 : dtype == HDB_F32 ? 4 : 8;
    a.row_bytes = a.d * elem;
    a.nchunks = a.row_bytes / 16;
    const bool vec = (a.row_bytes % 16 == 0) && ((reinterpret_cast<uintptr_t>(a.V) & 15) == 0) &&
                     ((size_t)a.d * (elem == 8 ? 8 : 4) <= 60 * 1024);
    // grid: every wave keeps a whole tile in flight.  With 12 KiB tiles (768-byte rows) 2 workgroups (8 waves) per CU
    // is the measured optimum (6.8 TB/s); with the 24 KiB tiles of the unrolled 1536-byte-row variant one workgroup
    // per CU is (N=1M fp32 d=384: 237 vs 246 us; N=10M: 7.02 vs 6.78 TB/s) -- the same ~100 KiB in flight per CU
    const bool wide_rows = vec && a.row_bytes == 6 * 256 && nq_launch < 2;      // (the one-query kernel; two or more queries take the four-query one)
    // (several query groups, grid.y > 1: fewer blocks along x -- to cut the flush atomics, one per block and query -- was measured:
    // n = 200k x 100 float32, 16 queries 143 -> 120 us, but 1M rows 418 -> 566 us: the groups drift apart and stop sharing V in L2)
    const int auto_blocks = wide_rows ? 256 : 512;
    const int blocks = hdb_grid_for(a.ntiles, 4, max_blocks > 0 ? max_blocks : auto_blocks);
    if (dtype == HDB_F16) { if (mode == 0) launch_scan_t<__half, 0>(a, nq_launch, blocks, vec, st); else launch_scan_t<__half, 1>(a, nq_launch, blocks, vec, st); }
    else if (dtype == HDB_F32) { if (mode == 0) launch_scan_t<float, 0>(a, nq_launch, blocks, vec, st); else launch_scan_t<float, 1>(a, nq_launch, blocks, vec, st); }
    else { if (mode == 0) launch_scan_t<double, 0>(a, nq_launch, blocks, vec, st); else launch_scan_t<double, 1>(a, nq_launch, blocks, vec, st); }
    return (int)hipGetLastError();
}

extern "C" int hdb_launch_rownorm(const void* V, int64_t n, int d, int dtype, float* inv_norm, float* sqnorm,
                                  int* nan_flag, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int blocks = hdb_grid_for((n + 15) / 16, 4, 2048);
    if (dtype == HDB_F16) hipLaunchKernelGGL(hdb_rownorm_kernel<__half>, dim3(blocks), dim3(256), 0, st, (const __half*)V, n, d, inv_norm, sqnorm, nan_flag);
    else if (dtype == HDB_F32) hipLaunchKernelGGL(hdb_rownorm_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)V, n, d, inv_norm, sqnorm, nan_flag);
    else hipLaunchKernelGGL(hdb_rownorm_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double*)V, n, d, inv_norm, sqnorm, nan_flag);
    return (int)hipGetLastError();
}

extern "C" int hdb_launch_qprep2(const void* Q, int nq, int d, bool f64, float* qinv, float* qsq, int* qnan, void* q16, float* qscl,
                                 float* thr_init, uint32_t* cnt_init, uint32_t* qbits, int W, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (f64) hipLaunchKernelGGL(hdb_qprep_kernel<double>, dim3(nq), dim3(64), 0, st, (const double*)Q, nq, d, qinv, qsq, qnan, (_Float16*)q16, qscl, thr_init, cnt_init, qbits, W);
    else hipLaunchKernelGGL(hdb_qprep_kernel<float>, dim3(nq), dim3(64), 0, st, (const float*)Q, nq, d, qinv, qsq, qnan, (_Float16*)q16, qscl, thr_init, cnt_init, qbits, W);
    return (int)hipGetLastError();
}
extern "C" int hdb_launch_qprep(const void* Q, int nq, int d, bool f64, float* qinv, float* qsq, int* qnan, void* q16, float* qscl,
                                void* stream) {
    return hdb_launch_qprep2(Q, nq, d, f64, qinv, qsq, qnan, q16, qscl, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int hdb_launch_signpack(const void* V, int64_t n, int d, int dtype, int64_t row0, uint32_t* bits, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (d % 32 == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0) {
        const int blocks = hdb_grid_for((n + 7) / 8, 4, 4096);      // a wave per eight rows
        if (dtype == HDB_F16) hipLaunchKernelGGL(hdb_signpack_wide_kernel<__half>, dim3(blocks), dim3(256), 0, st, (const __half*)V, n, d, row0, bits);
        else if (dtype == HDB_F32) hipLaunchKernelGGL(hdb_signpack_wide_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)V, n, d, row0, bits);
        else hipLaunchKernelGGL(hdb_signpack_wide_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double*)V, n, d, row0, bits);
        return (int)hipGetLastError();
    }
    const int blocks = hdb_grid_for(n, 4, 4096);
    if (dtype == HDB_F16) hipLaunchKernelGGL(hdb_signpack_kernel<__half>, dim3(blocks), dim3(256), 0, st, (const __half*)V, n, d, row0, bits);
    else if (dtype == HDB_F32) hipLaunchKernelGGL(hdb_signpack_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)V, n, d, row0, bits);
    else hipLaunchKernelGGL(hdb_signpack_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double*)V, n, d, row0, bits);
    return (int)hipGetLastError();
}

extern "C" int hdb_launch_qsign(const void* Q, int nq, int d, bool f64, int W, uint32_t* qbits, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (f64) hipLaunchKernelGGL(hdb_qsign_kernel<double>, dim3(nq), dim3(64), 0, st, (const double*)Q, nq, d, W, qbits);
    else hipLaunchKernelGGL(hdb_qsign_kernel<float>, dim3(nq), dim3(64), 0, st, (const float*)Q, nq, d, W, qbits);
    return (int)hipGetLastError();
}

extern "C" int hdb_launch_hamming(const ScanArgs* args, int mode, int nq_launch, const uint32_t* bits, int64_t npad,
                                  int W, const uint32_t* qbits, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const ScanArgs& a = *args;
    const int nq_end = a.q0 + nq_launch;
    const bool jac = a.metric == HDB_JACCARD;
    const int blocks = hdb_grid_for(a.ntiles * 4, 256, 2048);
#define HDB_HAM_LAUNCH(MODE_, JAC_, QH_)                                                                              \
    hipLaunchKernelGGL((hdb_hamming_kernel<MODE_, JAC_, QH_>), dim3(blocks, (nq_launch + QH_ - 1) / QH_), dim3(256), 0, st, a, bits, \
                       npad, W, qbits, nq_end)
    if (nq_launch == 1) {
        if (mode == 0) { if (jac) HDB_HAM_LAUNCH(0, true, 1); else HDB_HAM_LAUNCH(0, false, 1); }
        else { if (jac) HDB_HAM_LAUNCH(1, true, 1); else HDB_HAM_LAUNCH(1, false, 1); }
    } else {
        if (mode == 0) { if (jac) HDB_HAM_LAUNCH(0, true, 4); else HDB_HAM_LAUNCH(0, false, 4); }
        else { if (jac) HDB_HAM_LAUNCH(1, true, 4); else HDB_HAM_LAUNCH(1, false, 4); }
    }
#undef HDB_HAM_LAUNCH
    return (int)hipGetLastError();
}

// Row mask folded into a bias vector for the MFMA scan (which has no mask input): out = mask ? bias : -inf.
__global__ __launch_bounds__(256) void hdb_maskbias_kernel(const uint8_t* mask, const float* bias, int64_t n, float* out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = mask[i] ? (bias ? bias[i] : 0.f) : -INFINITY;
}
extern "C" int hdb_launch_maskbias(const uint8_t* mask, const float* bias, int64_t n, float* out, void* stream) {
    hipLaunchKernelGGL(hdb_maskbias_kernel, dim3(hdb_grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, mask, bias, n, out);
    return (int)hipGetLastError();
}

// Recency term of hyperDB_ranking_algorithm_sort (ranking_algorithm.py:180-183):
// bias[i] = recency_bias * exp(ts[i] - max(ts)), difference and exp in float64 (unix-second
// timestamps lose ~100 s of resolution in float32), result stored as float32.
__global__ __launch_bounds__(256) void hdb_recency_kernel(const double* ts, int64_t n, double rb, double ts_max, float* out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = (float)(rb * exp(ts[i] - ts_max));
}
extern "C" int hdb_launch_recency(const double* ts, int64_t n, double rb, double ts_max, float* out, void* stream) {
    hipLaunchKernelGGL(hdb_recency_kernel, dim3(hdb_grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, ts, n, rb, ts_max, out);
    return (int)hipGetLastError();
}

// Both decays of a HyperDB.query() call in one pass (reference hyperdb.py:1344: first = rb * exp(-max(ts) + ts) over the
// FILTERED documents, then ranking_algorithm.py:183: rb * exp(first - max(first))): the maxima are over the kept rows, which
// the caller knows as two scalars (first is monotone in ts: its maximum sits at the newest kept row for rb > 0 and at the
// oldest one for rb < 0).  float64 arithmetic, float32 result; rows the mask drops get 0 (they are never returned).
__global__ __launch_bounds__(256) void hdb_recency2_kernel(const double* ts, const uint8_t* mask, int64_t n, double rb, double ts_max,
                                                           double first_max, float* out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const bool keep = mask == nullptr || mask[i] != 0;
        const double first = rb * exp(-ts_max + ts[i]);
        out[i] = keep ? (float)(rb * exp(first - first_max)) : 0.f;
    }
}
extern "C" int hdb_launch_recency2(const double* ts, const uint8_t* mask, int64_t n, double rb, double ts_max, double first_max, float* out, void* stream) {
    hipLaunchKernelGGL(hdb_recency2_kernel, dim3(hdb_grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, ts, mask, n, rb, ts_max, first_max, out);
    return (int)hipGetLastError();
}

// Pearson (ranking_algorithm.py:77-113): r = sum((v-mv)(q-mq)) / (sd_v * sd_q * d).  Because sum(q-mq) = 0 the
// numerator is the plain dot product of v with the CENTRED query, so pearson runs on the cosine pipeline with
// two substitutions: per-row scale 1/(sd_v*d) instead of 1/||v|| (NaN for a constant row, like the reference's
// NaN rule :107-111) and per-query scale 1/sd_q (NaN for a constant query).  Row statistics are two-pass
// (mean, then centred squares) to avoid the cancellation of E[v^2]-E[v]^2.
template <typename T>
__global__ __launch_bounds__(256) void hdb_rowstats_kernel(const T* V, int64_t n, int d, float* pscale) {
    using Acc = typename Elem<T>::Acc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int64_t ntiles = (n + 15) / 16;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t r0 = t * 16 + 4 * g;
        Acc acc[4] = {Acc(0), Acc(0), Acc(0), Acc(0)};
        for (int e = l16; e < d; e += 16)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += (Acc)hdb_to_f(V[min(r0 + u, n - 1) * (int64_t)d + e]);
        const Acc mean = hdb_rows4_sum(acc[0], acc[1], acc[2], acc[3], l16) / Acc(d);   // mean of the OWNED row
        const int uo = hdb_owned_row(l16);
        // every lane needs the mean of each of its 4 rows: broadcast from the owning lanes of the group
        Acc mu[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int src = (lane & 48) + ((u & 1) * 8 + (u >> 1) * 4);     // lane of this group that owns row u
            mu[u] = __shfl(mean, src, 64);
        }
        Acc var[4] = {Acc(0), Acc(0), Acc(0), Acc(0)};
        for (int e = l16; e < d; e += 16)
#pragma unroll
            for (int u = 0; u < 4; ++u) { const Acc df = (Acc)hdb_to_f(V[min(r0 + u, n - 1) * (int64_t)d + e]) - mu[u]; var[u] += df * df; }
        const Acc ss = hdb_rows4_sum(var[0], var[1], var[2], var[3], l16);
        const int64_t row = r0 + uo;
        if ((l16 & 3) == 0 && row < n) {
            const Acc sd = sqrt(ss / Acc(d));
            pscale[row] = (sd == Acc(0)) ? NAN : (float)(Acc(1) / (sd * Acc(d)));
        }
    }
}

template <typename Acc>
__global__ __launch_bounds__(64) void hdb_qcentre_kernel(const Acc* Q, int nq, int d, Acc* Qc, float* qscale) {
    const int q = blockIdx.x;
    Acc s = Acc(0);
    for (int e = threadIdx.x; e < d; e += 64) s += Q[(int64_t)q * d + e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const Acc mean = s / Acc(d);
    Acc v = Acc(0);
    for (int e = threadIdx.x; e < d; e += 64) { const Acc c = Q[(int64_t)q * d + e] - mean; Qc[(int64_t)q * d + e] = c; v = fma(c, c, v); }   // (explicit: hdb_mfma_fused.h repeats it)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (threadIdx.x == 0) { const Acc sd = sqrt(v / Acc(d)); qscale[q] = (sd == Acc(0)) ? NAN : (float)(Acc(1) / sd); }
}

extern "C" int hdb_launch_rowstats(const void* V, int64_t n, int d, int dtype, float* pscale, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int blocks = hdb_grid_for((n + 15) / 16, 4, 2048);
    if (dtype == HDB_F16) hipLaunchKernelGGL(hdb_rowstats_kernel<__half>, dim3(blocks), dim3(256), 0, st, (const __half*)V, n, d, pscale);
    else if (dtype == HDB_F32) hipLaunchKernelGGL(hdb_rowstats_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)V, n, d, pscale);
    else hipLaunchKernelGGL(hdb_rowstats_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double*)V, n, d, pscale);
    return (int)hipGetLastError();
}
extern "C" int hdb_launch_qcentre(const void* Q, int nq, int d, bool f64, void* Qc, float* qscale, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (f64) hipLaunchKernelGGL(hdb_qcentre_kernel<double>, dim3(nq), dim3(64), 0, st, (const double*)Q, nq, d, (double*)Qc, qscale);
    else hipLaunchKernelGGL(hdb_qcentre_kernel<float>, dim3(nq), dim3(64), 0, st, (const float*)Q, nq, d, (float*)Qc, qscale);
    return (int)hipGetLastError();
}
