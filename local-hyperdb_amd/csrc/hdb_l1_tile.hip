// hdb_l1_tile.hip -- manhattan_distance batches (reference hyperdb/ranking_algorithm.py:54-61: 1 / (1 + sum |v - q|)) as ONE
// HBM pass per 8-64 queries.
//
// sum |v - q| has no matrix form, and the 4-query VALU scan (hdb_scan.hip) is ALU- and LDS-bound on it: every lane re-reads its
// query chunks from LDS for every row chunk and spends ~2.5 instructions per element and query (fp16, 5 M x 384: 636 us for one
// query, 2 560 us for five).  Here the tiles of V stream through the 3-deep LDS ring of the MFMA kernels (LDS-DMA by waves 4-7,
// source-side XOR swizzle) exactly once, the queries live in REGISTERS, and the work of a tile is cut by ROWS: wave w takes the
// 16-row pass w % NP of every tile for query group w / NP (NQH queries), so all four SIMDs share any batch size evenly.
//   float32 arithmetic for both data types (v_sub_f32 + v_add_f32 |x| per element and query; fp16 rows are converted once per
//   chunk and shared by the wave's queries): the queries are float32 -- a query that is not an fp16 value must not be rounded
//   to one (near-duplicate rows: the rounding of q, 5e-4 |q| per element, is a large share of a small |v - q|; packed fp16
//   differences were tried and failed the 1e-3 contract there) -- and the sums are bit-identical with the 4-query scan's.
// Row sums: the 5-step DPP ownership butterfly of hdb_common.h (4 rows per 16-lane group).  Epilogue as hdb_emit: 1 / (1 + sum),
// bias, NaN -> -inf, then scores (MODE 0) or threshold filter into the candidate lists (MODE 1).  Excluded rows arrive as a
// bias of -inf (hdb_maskbias_kernel), like in the MFMA scans.
#include "hdb_mfma_kernel.h"

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4l __attribute__((ext_vector_type(4)));

template <typename E, int D, int R, int NQH, int MODE, bool HAS_BIAS>
__global__ __launch_bounds__(512) void hdb_l1_tile_kernel(ScanArgs a, int nq_end) {
    constexpr int ES = (int)sizeof(E);
    constexpr int ROWB = D * ES, CPR = ROWB / 16, NJ = ROWB / 256, STAGE = R * ROWB;
    constexpr int NP = R / 16;                      // 16-row passes per tile
    constexpr int NG = 8 / NP;                      // query groups per workgroup
    constexpr int PPL = R * CPR / 256;              // 1-KiB LDS-DMA pieces per staging wave and tile
    static_assert(ROWB % 256 == 0 && (NP == 1 || NP == 2 || NP == 4) && STAGE <= 48 * 1024 && PPL + 1 <= 31, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bbuf = reinterpret_cast<float*>(smem + 3 * STAGE);       // [3][64] per-row bias of the staged tiles
    // survivors wait in LDS per query of the block and reach the global lists with one atomic per block and query (as in hdb_scan.hip:
    // a returning atomic per survivor on the per-query counters is what short passes spent their time in)
    constexpr int SCAP = 64;
    unsigned int* scnt = reinterpret_cast<unsigned int*>(smem + 3 * STAGE + 3 * 64 * 4 + 64);      // [NG * NQH] + [1] flush base
    unsigned long long* sbuf = reinterpret_cast<unsigned long long*>(smem + 3 * STAGE + 3 * 64 * 4 + 64 + 128);   // [NG * NQH][SCAP]
    if (MODE == 1 && threadIdx.x < NG * NQH + 1) scnt[threadIdx.x] = 0u;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g4 = lane >> 4;
    const int pass = w % NP, grp = w / NP;
    const bool stager = w >= 4;
    // the queries of this block (at most NG * NQH) are spread evenly over the NG query groups: five queries = 3 + 2, not 4 + 1
    // (the wave with the most queries sets the pace of its SIMD)
    const int qb0 = a.q0 + (int)blockIdx.y * (NG * NQH);
    const int nqb = nq_end - qb0 < NG * NQH ? nq_end - qb0 : NG * NQH;
    const int per = (nqb + NG - 1) / NG;                                       // queries per group (the last groups may have fewer)
    const int q0w = qb0 + grp * per;                                           // this wave's first query
    const int nqw = nqb - grp * per < per ? (nqb - grp * per > 0 ? nqb - grp * per : 0) : per;      // queries this wave really has
    const bool wave_active = nqw > 0;

    // ---- this wave's queries live in registers: lane (g4, l16) holds chunks l16 + 16 j of every query (E elements, as V stores them).
    // fp16 rows, round 4: when every element of this wave's queries IS an fp16 number (the normal case: queries embedded in the store's
    // precision) the differences are taken in packed fp16 -- v_pk_add_f16 (one rounding: v - q is exact for rows near the query, by
    // Sterbenz, and off by at most 2^-11 of itself elsewhere; an error of the SUM of ~2^-11 / sqrt(d), far inside the 1e-3 contract),
    // sign bits masked off, v_dot2c_f32_f16 against (1, 1) for the float32 sum: 3 instructions per TWO elements and query instead of
    // 2 per element plus the conversions, and half the registers per query.  A query that is not an fp16 number keeps the float32
    // path (its rounding would be a large share of a small |v - q|, see the header).  The choice is wave-uniform, made once, and
    // selects one of two copies of the tile loop (the query registers of the other flavour never exist).
    constexpr int EPC = 16 / ES;                    // elements per 16-byte chunk of V
    float thr[NQH];
#pragma unroll
    for (int q = 0; q < NQH; ++q) thr[q] = (MODE == 1 && q < nqw) ? a.thr[q0w + q - a.q0] : INFINITY;
    auto query_src = [&](int q) {
        const int qq = q < nqw ? q0w + q : nq_end - 1;
        return static_cast<const float*>(a.Q) + (int64_t)(qq < 0 ? 0 : qq) * D;
    };
    bool pk = false;
    if constexpr (ES == 2) {
        bool exact = wave_active && a.dyn_heavy != 77;        // (ScanArgs::dyn_heavy = 77: keep the float32 arithmetic -- set_option l1_packed 0, A/B runs)
        for (int q = 0; q < nqw; ++q) {
            const float* src = query_src(q);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < EPC; e += 4) {
                    const float4 x = *reinterpret_cast<const float4*>(src + (l16 + 16 * j) * EPC + e);
                    exact = exact && (float)(_Float16)x.x == x.x && (float)(_Float16)x.y == x.y && (float)(_Float16)x.z == x.z && (float)(_Float16)x.w == x.w;
                }
        }
        pk = __builtin_amdgcn_readfirstlane((int)__all(exact)) != 0;
    }

    const char* const Vb = reinterpret_cast<const char*>(a.V);
    const int64_t n_rows = a.n;
    const int64_t ntiles = (n_rows + R - 1) / R;
    const int64_t G = gridDim.x, bidx = blockIdx.x;
    auto issue = [&](int64_t t, int st) {                 // waves 4-7: the whole tile (and its bias), rows past the end repeat the last row
        if (!stager) return;
        const int64_t row0 = t * R;
        const int64_t last = n_rows - 1 - row0;
        char* sdst = smem + st * STAGE;
        const char* tile_base = Vb + row0 * (int64_t)ROWB;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int pc = (w & 3) + 4 * j;
            const int slot = pc * 64 + lane;
            const int r = slot / CPR, cpos = slot - r * CPR;
            const int rr = r <= (int)last ? r : (int)last;
            const unsigned int off = (unsigned int)(rr * ROWB + (cpos ^ (r & 15)) * 16);
            __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(tile_base + off), HDB_LDS_PTR(sdst + pc * 1024), 16, 0, 2);
        }
        if (HAS_BIAS) {
            const int64_t rr = lane <= last ? lane : last;
            __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(a.bias + row0 + rr), HDB_LDS_PTR(bbuf + st * 64), 4, 0, 0);
        }
    };
    constexpr int NLOAD = PPL + (HAS_BIAS ? 1 : 0);
    int64_t tA = bidx, tB = bidx + G;
    if (tA < ntiles) issue(tA, 0);
    if (tB < ntiles) issue(tB, 1);
    const unsigned int smem_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(smem);
    const int u_own = hdb_owned_row(l16);
    int st_cur = 0;
    auto tile_loop = [&](auto flavour) {
    constexpr bool PK = decltype(flavour)::value != 0;
    float qv[PK ? 1 : NQH][PK ? 1 : NJ][EPC];
    unsigned int qpk[PK ? NQH : 1][PK ? NJ : 1][4];
#pragma unroll
    for (int q = 0; q < NQH; ++q) {
        const float* src = query_src(q);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int e = 0; e < EPC; e += 4) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (wave_active) x = *reinterpret_cast<const float4*>(src + (l16 + 16 * j) * EPC + e);
                if constexpr (PK) {
                    const half2v h0 = {(_Float16)x.x, (_Float16)x.y}, h1 = {(_Float16)x.z, (_Float16)x.w};
                    qpk[q][j][e / 2] = __builtin_bit_cast(unsigned int, h0); qpk[q][j][e / 2 + 1] = __builtin_bit_cast(unsigned int, h1);
                } else {
                    qv[q][j][e] = x.x; qv[q][j][e + 1] = x.y; qv[q][j][e + 2] = x.z; qv[q][j][e + 3] = x.w;
                }
            }
        }
    }
    for (; tA < ntiles; tA = tB, tB += G) {
        if (stager) { if (tB < ntiles) hdb_wait_vmcnt<NLOAD>(); else hdb_wait_vmcnt<0>(); }
        hdb_lds_barrier();                                // tile tA is in LDS; everyone is done with the tile before it
        const int st_next2 = st_cur == 0 ? 2 : st_cur - 1;
        if (tB + G < ntiles) issue(tB + G, st_next2);
        if (wave_active) {
            float acc[4][NQH];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < NQH; ++q) acc[u][q] = 0.f;
            const unsigned int tb = smem_addr + (unsigned int)(st_cur * STAGE);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                u32x4l raw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {               // rows 16 pass + 4 g4 + u; chunk c of row r sits at ((c ^ (r & 15)) << 4) of its row image
                    const int r = 16 * pass + 4 * g4 + u;
                    const unsigned int ad = tb + (unsigned int)(r * ROWB) + (unsigned int)(((l16 + 16 * j) ^ (r & 15)) << 4);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(raw[u]) : "v"(ad));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]));
                if constexpr (PK) {
                    const half2v ones = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        // (one cast of the whole chunk into a struct of four dwords: __builtin_bit_cast of an ELEMENT of an ext_vector, raw[u][dw],
                        //  reads element 0 whatever the index (hipcc 7.2) -- the float32 flavour casts the whole half8 for the same reason)
                        const uint4 r4 = __builtin_bit_cast(uint4, raw[u]);
                        const unsigned int rd[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                        for (int q = 0; q < NQH; ++q)
                            if (q < nqw) {
#pragma unroll
                                for (int dw = 0; dw < 4; ++dw) {
                                    const half2v df = __builtin_bit_cast(half2v, rd[dw]) - __builtin_bit_cast(half2v, qpk[q][j][dw]);
                                    const unsigned int ab = __builtin_bit_cast(unsigned int, df) & 0x7FFF7FFFu;
                                    acc[u][q] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, ab), ones, acc[u][q], false);
                                }
                            }
                    }
                } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float x[EPC];
                    if constexpr (ES == 2) {
                        const half8 hv = __builtin_bit_cast(half8, raw[u]);       // (one cast of the whole chunk: element-wise casts of raw[u][e] all read dword 0)
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = (float)hv[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = __uint_as_float(raw[u][e]);
                    }
#pragma unroll
                    for (int q = 0; q < NQH; ++q)
                        if (q < nqw) {                   // (wave-uniform: slots beyond the batch cost a scalar branch, not the arithmetic;
                                                         //  two unrolled copies of the body -- all slots / some -- spilled the queries)
#pragma unroll
                            for (int e = 0; e < EPC; ++e) acc[u][q] += fabsf(x[e] - qv[q][j][e]);
                        }
                }
                }
            }
            // ---- epilogue: one row per owning lane ((l16 & 3) == 0) and query
            const int rloc = 16 * pass + 4 * g4 + u_own;
            const int64_t row = tA * R + rloc;
            float b0 = 0.f;
            if (HAS_BIAS) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b0) : "v"((unsigned int)(uintptr_t)HDB_LDS_PTR(bbuf) + (unsigned int)(st_cur * 64 + rloc) * 4u) : "memory");
#pragma unroll
            for (int q = 0; q < NQH; ++q) {
                const float sum = hdb_rows4_sum(acc[0][q], acc[1][q], acc[2][q], acc[3][q], l16);
                float s = 1.0f / (1.0f + sum);                                    // reference :59-60
                if (HAS_BIAS) s += b0;
                s = hdb_canon(s);
                const int qg = q0w + q;
                if ((l16 & 3) == 0 && q < nqw && row < n_rows) {
                    const int ql = qg - a.q0;
                    if (MODE == 0) a.scores[(int64_t)ql * a.ld + row] = s;
                    else if (s >= thr[q] && !(HAS_BIAS && s == -INFINITY)) {
                        const unsigned long long ent = hdb_pack(s, (uint32_t)row);
                        const int slot = qg - qb0;
                        const unsigned int lp = atomicAdd(&scnt[slot], 1u);             // LDS
                        if (lp < (unsigned int)SCAP) sbuf[slot * SCAP + (int)lp] = ent;
                        else {
                            const uint32_t pos = atomicAdd(&a.cnt[ql * HDB_CNT_STRIDE], 1u);
                            if (pos < a.cap) a.cand[(int64_t)ql * a.cap + pos] = ent;
                        }
                    }
                }
            }
        }
        st_cur = st_cur == 2 ? 0 : st_cur + 1;
    }
    };
    if constexpr (ES == 2) { if (pk) tile_loop(HdbIC<1>()); else tile_loop(HdbIC<0>()); }
    else tile_loop(HdbIC<0>());
    if (MODE == 1) {
        __syncthreads();
        for (int slot = 0; slot < nqb; ++slot) {
            const unsigned int have = scnt[slot] < (unsigned int)SCAP ? scnt[slot] : (unsigned int)SCAP;      // (block-uniform)
            if (have == 0u) continue;
            const int ql = qb0 - a.q0 + slot;
            if (tid == 0) scnt[NG * NQH] = atomicAdd(&a.cnt[ql * HDB_CNT_STRIDE], have);
            __syncthreads();
            const unsigned int base = scnt[NG * NQH];
            for (unsigned int e = (unsigned int)tid; e < have; e += 512u)
                if (base + e < a.cap) a.cand[(int64_t)ql * a.cap + base + e] = sbuf[slot * SCAP + (int)e];
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <typename E, int D, int R, int NQH, int MODE, bool HAS_BIAS>
static int l1_launch_one(const ScanArgs& a, int nq_launch, int blocks, hipStream_t st) {
    auto kern = hdb_l1_tile_kernel<E, D, R, NQH, MODE, HAS_BIAS>;
    const size_t lds = (size_t)3 * R * D * sizeof(E) + 3 * 64 * 4 + 64 + 128 + (size_t)(8 / (R / 16)) * NQH * 64 * 8;      // tiles, bias, survivor counters and slots
    static unsigned long long attr_done = 0;
    hipError_t e = hdb_lds_attr_once(reinterpret_cast<const void*>(kern), (int)lds, &attr_done);
    if (e != hipSuccess) return (int)e;
    constexpr int per_block = (8 / (R / 16)) * NQH;
    const dim3 grid(blocks, (nq_launch + per_block - 1) / per_block);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a, a.q0 + nq_launch);
    return (int)hipGetLastError();
}
template <typename E, int D, int R, int NQH>
static int l1_launch(const ScanArgs& a, int mode, int nq_launch, int blocks, hipStream_t st) {
    const bool b = a.bias != nullptr;
    if (mode == 0) return b ? l1_launch_one<E, D, R, NQH, 0, true>(a, nq_launch, blocks, st) : l1_launch_one<E, D, R, NQH, 0, false>(a, nq_launch, blocks, st);
    return b ? l1_launch_one<E, D, R, NQH, 1, true>(a, nq_launch, blocks, st) : l1_launch_one<E, D, R, NQH, 1, false>(a, nq_launch, blocks, st);
}

// rows that are multiples of 256 bytes up to 1536 bytes (fp16 d <= 768, float32 d <= 384: two or four float32 queries per wave in registers)
// Measured against the 4-query scan in one process (profiles/r3_manhattan_tile_vs_scan.txt): fp16 d=384, 5 M rows: 800 vs 1 372 us
// (2 queries), 1 187 vs 2 558 (5), 1 390 vs 2 728 (8); fp16 d=128: 559 vs 986 (5); float32 d=384, 2 M rows: 594 vs 1 083 (5).
// One query: equal (the single-query scan keeps it).  fp16 d = 512 keeps two queries per wave, d = 640 / 768 one (two copies of a
// 768-element query are 96 registers next to the tile chunks: spills); they still share the staged tile between the waves:
// d = 768, 2.5 M rows: 589 vs 1 231 us (2 queries), 1 531 vs 2 484 (5), 3 674 vs 4 813 (16); d = 512, 4 M rows: 1 275 vs 2 686 (5).
extern "C" int hdb_l1_tile_supported(int dtype, int d) {
    if (dtype == HDB_F16) return d == 128 || d == 256 || d == 384 || d == 512 || d == 640 || d == 768;      // (512: two queries per wave; 640 / 768: one)
    if (dtype == HDB_F32) return d == 128 || d == 256 || d == 384 || d == 512 || d == 768;      // (512: two queries per wave, 16-row tiles; 768: one)
    return 0;
}

// dense passes only (a.tile_stride == 1): the row sample stays with the 4-query scan.  The caller folds a row mask into the bias.
extern "C" int hdb_launch_l1_tile(const ScanArgs* args, int dtype, int mode, int nq_launch, int max_blocks, void* stream) {
    const ScanArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    if (a.mask || a.tile_stride != 1) return (int)hipErrorNotSupported;
    const int cus = hdb_cu_count();
    const int rowb = a.d * (dtype == HDB_F16 ? 2 : 4);
    const int R = rowb <= 768 ? 64 : rowb <= 1536 ? 32 : 16;
    const int64_t tiles = (a.n + R - 1) / R;
    int blocks = (int)(tiles < cus ? tiles : cus);
    if (max_blocks > 0 && max_blocks < blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    if (dtype == HDB_F16) {
        switch (a.d) {
            case 128: return l1_launch<_Float16, 128, 64, 4>(a, mode, nq_launch, blocks, st);
            case 256: return l1_launch<_Float16, 256, 64, 4>(a, mode, nq_launch, blocks, st);
            case 384: return l1_launch<_Float16, 384, 64, 4>(a, mode, nq_launch, blocks, st);
            case 512: return l1_launch<_Float16, 512, 32, 2>(a, mode, nq_launch, blocks, st);
            case 640: return l1_launch<_Float16, 640, 32, 1>(a, mode, nq_launch, blocks, st);
            case 768: return l1_launch<_Float16, 768, 32, 1>(a, mode, nq_launch, blocks, st);
            default: return (int)hipErrorNotSupported;
        }
    }
    switch (a.d) {
        case 128: return l1_launch<float, 128, 64, 4>(a, mode, nq_launch, blocks, st);
        case 256: return l1_launch<float, 256, 32, 4>(a, mode, nq_launch, blocks, st);
        case 384: return l1_launch<float, 384, 32, 2>(a, mode, nq_launch, blocks, st);      // (four queries spill three registers)
        case 512: return l1_launch<float, 512, 16, 2>(a, mode, nq_launch, blocks, st);
        case 768: return l1_launch<float, 768, 16, 1>(a, mode, nq_launch, blocks, st);
        default: return (int)hipErrorNotSupported;
    }
}
