// hdb_mfma_kernel.h -- the MFMA row-scan kernel template and its launch helpers, shared by the fp16 (hdb_mfma.hip)
// and fp32 (hdb_mfma_f32.hip) translation units.  See hdb_mfma.hip for the description of the kernel.
#pragma once
#include "hdb_common.h"
#include "../../include/hyperdb_hip.h"
#include "hdb_finalize.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HDB_MFMA_CB 1024            // LDS candidate list entries per workgroup
#define HDB_MFMA_SEG (HDB_MFMA_CB / 8)   // ... per wave
// Measurement builds only (tools/knockout_q256.py; the product is built with 0): 1 = survivors are never appended,
// 2 = no LDS-DMA once the ring is primed (stale but random tiles), 4 = no per-tile barrier, 8 = half of the fragment reads (row tiles 1 and 3 reuse 0 and 2),
// 16 = no in-place conversion of float32 tiles (hdb_f32s), 32 = one of the five part products only.  Results are wrong by design.
#ifndef HDB_MFMA_KNOCKOUT
#define HDB_MFMA_KNOCKOUT 0
#endif

// Diagnostic build only (tools/clock_q256.py; the product is built with 0): stamp s_memtime (shader clock) and
// s_memrealtime (constant 100 MHz) around the tile loop; lane 0 of wave 0 of every workgroup stores the four stamps
// into a buffer of their own that no kernel reads (MI355X_MICROARCH.md, DVFS give-back item 6).  The in-kernel clock
// is d(memtime) / d(memrealtime) x 100 MHz.
#ifndef HDB_MFMA_CLOCK
#define HDB_MFMA_CLOCK 0
#endif
// Diagnostic build only (tools/stamps_batch1.py; product: 0): wall-clock stamps (s_memrealtime, 100 MHz) of the phases of
// the single-launch batched call, per workgroup, in a buffer of their own that nothing reads: [wg][16] = 0 start,
// 1 queries prepared, 2 sample pass done, 3 published, 4 owner done, 5 thresholds known, 6 filter pass done, 7 arrived,
// 8 everybody arrived, 9 own queries sorted, 10 left.
#ifndef HDB_BATCH_STAMPS
#define HDB_BATCH_STAMPS 0
#endif
#if HDB_BATCH_STAMPS
static __device__ unsigned long long hdb_batch_stamps[16 * 1024];
#define HDB_BSTAMP(slot) do { if (ONE && tid == 0 && blockIdx.x < 1024) hdb_batch_stamps[16 * blockIdx.x + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HDB_BSTAMP(slot) do { } while (0)
#endif
// Diagnostic build only (tools/round_sections.py; product: 0): every wave adds up the shader clocks (s_memtime) it spends in the
// sections of a round -- [0] waiting at the round's barrier, [1] staging, [2] multiplying (+ handing partial sums over), [3] waiting
// at the K-part barrier, [4] epilogue, [5] waiting for / converting the next tile, [6] rounds -- and stores them per workgroup and
// wave into a buffer of its own at the end of the launch.
#ifndef HDB_ROUND_PROF
#define HDB_ROUND_PROF 0
#endif
#if HDB_ROUND_PROF
static __device__ unsigned long long hdb_round_prof[256 * 8 * 8];
#define HDB_RP(slot) do { const unsigned long long rp_now = __builtin_amdgcn_s_memtime(); rp_acc[slot] += rp_now - rp_t; rp_t = rp_now; } while (0)
#else
#define HDB_RP(slot) do { } while (0)
#endif
#if HDB_MFMA_CLOCK
#define HDB_CLOCK_WGS 1024
static __device__ unsigned long long hdb_clock_buf[4 * HDB_CLOCK_WGS];
#endif

// Staging roles (measured, profiles/r2_q256_experiments.json, N=10M d=384 Q=256, interleaved rounds per variant on one
// MI355X each): with every wave staging its eighth of a tile (round 1) the two waves of a SIMD both stalled on their own
// LDS-DMA issue (~100-185 blocked cycles per 1-KiB piece): 1.906 ms.  Variants: pieces spread between the MFMAs 1.935,
// fragment reads before the deferred epilogue 1.938, s_setprio 1 for waves 4-7 1.890 on one box and worse on another,
// for waves 0-3 1.906, waves 0-3 staging everything 1.803, only waves 6-7 staging 1.912, and the one that is shipped:
// WAVES 4-7 STAGE EVERY TILE (PPL pieces each) right after the barrier, waves 0-3 never issue LDS-DMA: 1.768 ms.
// The two waves of a SIMD then take turns on the matrix pipe -- A multiplies while B stages and filters, then B
// multiplies while A filters.  In the HBM-bound passes (up to 64 queries waves 4-7 do not multiply at all) the same
// split gives d=384 Q=8 1.18 -> 1.12 ms, Q=64 1.30 -> 1.21, d=768 Q=64 euclidean + bias 2.54 -> 2.41, d=128 Q=48
// 0.586 -> 0.489, N=1.25M Q=16 251 -> 214 us.
// Later A/B on the shipped roles (256 queries, one box, 1.805-1.81 ms that day): waves 0-3 taking 3, 4, 6 or 8 twelfths of
// the pieces and issuing them at the END of their round (after multiply and filter, while waves 4-7 multiply): 1.812,
// 1.806, 1.807, 1.810; all twelve twelfths: 1.795 (-0.8 %).  The schedule of the staging no longer moves the pass: with
// the staging knocked out altogether it takes 1.53 ms, with staging from L2-resident tiles 1.70 -- what is left is the
// energy of the stream itself (HBM read + LDS write) under the power limit, see DESIGN.md section 6.

#define HDB_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define HDB_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void hdb_wait_vmcnt() {      // s_waitcnt takes an immediate: one asm per value
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    else if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if constexpr (N == 23) asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 25) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
    else if constexpr (N == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
    else if constexpr (N == 27) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
    else if constexpr (N == 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    else if constexpr (N == 29) asm volatile("s_waitcnt vmcnt(29)" ::: "memory");
    else if constexpr (N == 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
    else if constexpr (N == 31) asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
    else static_assert(N < 0, "vmcnt immediate out of range");
}
__device__ __forceinline__ void hdb_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// LDS accesses in inline asm: hipcc cannot prove them disjoint from the ring that LDS-DMA writes and would drain the
// wave's in-flight staging (s_waitcnt vmcnt(0)) in front of every one of them.
__device__ __forceinline__ void hdb_lds_st32(unsigned int addr, float v) {
    asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void hdb_lds_st16(unsigned int addr, unsigned int v) {
    asm volatile("ds_write_b16 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void hdb_lds_st128(unsigned int addr, f32x4 v) {
    asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ float hdb_lds_ld32(unsigned int addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// 16-byte agent-scope (sc1) store and a pair of 16-byte sc1 loads: the exchange granules {epoch, key} travel two at a time
// (an 8-byte sc1 store is one fabric write of its own; four lanes storing 16 bytes each fill a 64-byte line in one)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void hdb_st128_sc1(const void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void hdb_ld128x2_sc1(const void* p0, const void* p1, u32x4& a, u32x4& b) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(p0), "v"(p1) : "memory");
}
typedef __attribute__((address_space(1))) unsigned long long hdb_gu64;
typedef __attribute__((address_space(1))) unsigned int hdb_gu32;


template <int V> struct HdbIC { static constexpr int value = V; };

// Element tag of the float32 row scan that multiplies in THREE bf16 PARTS (below): the bytes in memory are plain float32
struct hdb_f32s { float x; __device__ __forceinline__ operator float() const { return x; } };
typedef __bf16 hdb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct HdbRaw8 { f32x4 lo, hi; };                     // 8 floats of a row (two 16-byte chunks)
struct HdbParts3 { u32x4 p0, p1, p2; };               // the same 8 values as three bf16x8 fragments, v = p0 + p1 + p2 EXACTLY
// Truncation split: p0 = the upper 16 bits of v (sign, exponent, 7 mantissa bits), r = v - p0 (exact: at most 16 significant bits
// left), p1 = the upper 16 bits of r, p2 = r - p1 (at most 8 significant bits: a bf16 as it stands).  bf16 has float32's exponent
// range, so nothing under- or overflows on the way that float32 itself would not.  9 VALU instructions per pair of values.
// (hipcc 7.2: __builtin_bit_cast applied to an ELEMENT of an ext_vector reads element 0 whatever the index -- the bits go through
// by-value scalars here)
__device__ __forceinline__ unsigned int hdb_fbits(float f) { return __builtin_bit_cast(unsigned int, f); }
__device__ __forceinline__ float hdb_bitsf(unsigned int u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ HdbParts3 hdb_split3(const HdbRaw8& r) {
    HdbParts3 o;
    const float x[8] = {r.lo[0], r.lo[1], r.lo[2], r.lo[3], r.hi[0], r.hi[1], r.hi[2], r.hi[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v0 = x[2 * j], v1 = x[2 * j + 1];
        const unsigned int u0 = hdb_fbits(v0), u1 = hdb_fbits(v1);
        const f32x2 v = {v0, v1};
        const f32x2 h = {hdb_bitsf(u0 & 0xFFFF0000u), hdb_bitsf(u1 & 0xFFFF0000u)};
        const f32x2 r1 = v - h;
        const float r10 = r1[0], r11 = r1[1];
        const unsigned int w0 = hdb_fbits(r10), w1 = hdb_fbits(r11);
        const f32x2 g = {hdb_bitsf(w0 & 0xFFFF0000u), hdb_bitsf(w1 & 0xFFFF0000u)};
        const f32x2 r2 = r1 - g;
        const float r20 = r2[0], r21 = r2[1];
        o.p0[j] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                    // {upper half of u1, upper half of u0}
        o.p1[j] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
        o.p2[j] = __builtin_amdgcn_perm(hdb_fbits(r21), hdb_fbits(r20), 0x07060302u);
    }
    return o;
}

// MF: rows / queries per MFMA tile; E: element type of V and of the query fragments.  CPS = 16-byte chunks per k-step
// (one ds_read_b128 per lane and k-step: lane group h = lane / MF holds chunk CPS*s + h of its row).
// Eight floats -> [bf16(v) x 8][bf16(v - bf16(v)) x 8], round to nearest even; element order = the order hdb_split3 packs in
typedef __bf16 hdb_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void hdb_round2(const f32x4& lo, const f32x4& hi, u32x4& a0, u32x4& a1) {
    const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 v = {x[2 * j], x[2 * j + 1]};
        const hdb_bf16x2 b0 = __builtin_convertvector(v, hdb_bf16x2);
        const unsigned int P0 = __builtin_bit_cast(unsigned int, b0);
        const f32x2 h = {hdb_bitsf(P0 << 16), hdb_bitsf(P0 & 0xFFFF0000u)};
        const f32x2 r = v - h;
        const hdb_bf16x2 b1 = __builtin_convertvector(r, hdb_bf16x2);
        a0[j] = P0; a1[j] = __builtin_bit_cast(unsigned int, b1);
    }
}
template <int MF, typename E> struct MfmaShape;
template <> struct MfmaShape<32, _Float16> {
    using Acc = f32x16; using Vec = half8; using BVec = Vec;
    static constexpr int RPF = 1;
    static constexpr bool CONV = false;
    static constexpr int CPS = 2, NGRP = 4;          // groups of 4 consecutive rows per lane per tile
    __device__ static __forceinline__ Acc mma(Vec a, Vec b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaShape<16, _Float16> {
    using Acc = f32x4; using Vec = half8; using BVec = Vec;
    static constexpr int RPF = 1;
    static constexpr bool CONV = false;
    static constexpr int CPS = 4, NGRP = 1;
    __device__ static __forceinline__ Acc mma(Vec a, Vec b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
// fp32 data: v_mfma_f32_16x16x4_f32 takes ONE float of A and of B per lane (k slot = lane / 16).  The 16-byte
// fragment a lane reads holds 4 consecutive floats of its chunk, used as the k slots of four chained MFMAs: the
// assignment of actual k indices to (MFMA, slot) pairs is a permutation that A and B share, which is all a sum needs.
template <> struct MfmaShape<16, float> {
    using Acc = f32x4; using Vec = f32x4; using BVec = Vec;
    static constexpr int RPF = 1;
    static constexpr bool CONV = false;
    static constexpr int CPS = 4, NGRP = 1;
    __device__ static __forceinline__ Acc mma(Vec a, Vec b, Acc c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    }
};
// float32 data on the bf16 pipe.  A row value travels as TWO bf16 parts, a0 = bf16(v) and a1 = bf16(v - a0) (round to nearest even,
// v_cvt_pk_bf16_f32): |v - a0 - a1| <= 2^-18 |v|.  The workgroup converts every staged tile IN PLACE, once, whatever the number of
// queries (hdb_mfma_kernel's convert_tile): the 8 floats of a 32-byte group become [a0 x 8][a1 x 8], so the lane that owns those 8
// k slots still reads its two 16-byte chunks and finds one bf16x8 fragment in each.  A query value is split once per call into
// THREE parts that add up to it exactly (hdb_split3).  v . q = a0 q0 + a0 q1 + a1 q0 + a0 q2 + a1 q1 (the dropped a1 q2 is below
// 2^-26 |v||q|): five v_mfma_f32_16x16x32_bf16 = 80 cycles for 16 x 16 x 32 products against 8 x 32 cycles of v_mfma_f32_16x16x4_f32.
// Error of a score: at most 2^-18 = 3.8e-6 of |v||q| if every rounding pointed the same way; on real rows 1-2e-7, what float32
// accumulation itself leaves (tools/time_f32_split.py).
template <> struct MfmaShape<16, hdb_f32s> {
    using Acc = f32x4; using Vec = HdbRaw8; using BVec = HdbParts3;
    static constexpr int CPS = 8, NGRP = 1, RPF = 2;
    static constexpr bool CONV = true;
    __device__ static __forceinline__ Acc mma(const Vec& a, const HdbParts3& b, Acc c) {
#define HDB_BF(x) __builtin_bit_cast(hdb_bf16x8, x)
        if (!(HDB_MFMA_KNOCKOUT & 32)) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HDB_BF(a.hi), HDB_BF(b.p1), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HDB_BF(a.lo), HDB_BF(b.p2), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HDB_BF(a.hi), HDB_BF(b.p0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HDB_BF(a.lo), HDB_BF(b.p1), c, 0, 0, 0);
        }
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(HDB_BF(a.lo), HDB_BF(b.p0), c, 0, 0, 0);
#undef HDB_BF
    }
};

// METRIC: 0 dot, 1 cosine (aux0 = 1/||v||), 2 euclidean similarity (aux0 = ||v||^2)
// Fragment maps (lane l):  MF=32: row/query l&31, k = 16s + 8(l>>5) + j, C reg e -> row (e&3) + 8(e>>2) + 4(l>>5)
//                          MF=16: row/query l&15, k = 32s + 8(l>>4) + j, C reg e -> row 4(l>>4) + e
// MODE: 0 = store the scores (a.scores), 1 = filter against a.thr into the candidate lists, 2 = THE WHOLE CALL IN ONE LAUNCH
// for up to (8 / RS) * MF * QT queries (grid.y == 1, one persistent workgroup per CU):
//   prologue  every wave prepares its own queries from the caller's float32 vectors (hdb_qprep_kernel's sums in the same
//             order, so 1/||q|| and ||q||^2 are bit-identical; power-of-two scaled fp16 copies straight into the B fragments);
//   phase A   workgroup b multiplies tiles b, b+G, ... of the strided, jittered row sample (the plan of the multi-kernel
//             pipeline); every lane keeps the TWO largest comparable values it has seen per query tile (a lane sees
//             16 rows x RT of a tile for one query: with ~5 sample tiles per workgroup three of a query's top 8 land on one
//             of its 1024 lanes with probability 5e-5, and then only make the threshold a little less selective);
//   exchange  the lanes publish their two values as {epoch, key} granules [query][workgroup][8]; the OWNER of query q
//             (workgroup q mod G) sweeps that query's G x 8 granules until all carry this call's epoch, takes the 8-th
//             largest (a lower bound of the 8-th largest sample score: order statistic of a subset) and publishes it as
//             one {epoch, key} word; every workgroup polls the nq words.  The first two filter tiles are in flight meanwhile;
//   phase B   the filter pass over all rows (MODE 1's loop); survivors go to the global per-query lists;
//   finish    drain, agent-scope release, arrive; when every workgroup has arrived, the owner of each query sorts its
//             list in the (now free) ring LDS and writes its k results and its status word (euclidean: near-duplicates
//             are re-scored directly first, hdb_rescore_euclid_kernel's arithmetic); the last workgroup to leave zeroes
//             the control block.
// Every spin is bounded (s_memrealtime); a workgroup that gives up raises the abort word, which turns the statuses into
// HDB_Q_UNDERFLOW so that the host re-runs those queries through the exact path.
// NW: waves per workgroup.  8 is the product.  4 (MODE 1 only; tools/exp_4wave.py, profiles/r3_q256_four_waves.txt) is the
// measurement variant "one wave per SIMD with up to 512 registers, 64 queries per wave": every wave stages a quarter of each tile
// AND multiplies, the LDS fragment traffic per MFMA halves.
// KSL: the launch covers ONE K slice of D elements of rows that are wider (ScanArgs::ks_*): strided row and query addressing,
// accumulators start from the partial sums of the slices before; MODE 3 (KSL only) stores the raw sums for the next slice.
// KP = 2 ("K parts", float32 rows as bf16 parts only): waves w and w + 4 share a query group and take one half of the k-steps each,
// so a wave holds the query fragments of HALF a row (d = 512 / 768: 96 / 144 registers instead of 192 / 288) and all four SIMDs
// multiply even for 16 queries.  Wave w + 4 hands its 16 x 16 partial sums over through LDS (its own, otherwise unused, segment of
// the candidate list: 64 lanes x 16 bytes) behind a second barrier per tile; wave w adds them and runs the epilogue.  64 queries
// per launch row (grid.y) instead of 128.
template <typename E, int MF, int QT, int D, int R, int RS, int MODE, int METRIC, bool HAS_BIAS, int NW = 8, bool KSL = false, int KP = 1>
__global__ __launch_bounds__(NW * 64) void hdb_mfma_kernel(ScanArgs a, const E* __restrict__ q16,
                                                       const float* __restrict__ aux0g, const float* __restrict__ qsq, const float* __restrict__ qscl,
                                                       int nq_end, BatchArgs f) {
    using Shape = MfmaShape<MF, E>;
    using Vec = typename Shape::Vec;
    constexpr int ES = (int)sizeof(E);          // bytes per element
    constexpr int ROWB = D * ES;                // bytes per row
    using Acc = typename Shape::Acc;
    constexpr int NGRP = Shape::NGRP;
    constexpr int CPR = ROWB / 16;              // 16-byte chunks per row
    constexpr int CPS = Shape::CPS;             // chunks per k-step (2, 4 or 8)
    constexpr int RPF = Shape::RPF;             // 16-byte chunks a lane reads per k-step and row tile (2: float32 rows in bf16 parts)
    using BVec = typename Shape::BVec;
    constexpr bool CONV = Shape::CONV;          // float32 tiles are turned into bf16 parts in place before they are multiplied
    static_assert(CPR % CPS == 0 && (RPF == 1 || MF == 16), "k-steps");
    constexpr int KS = CPR / CPS / KP;          // k-steps of this wave (one fragment read each)
    static_assert(KP == 1 || (KP == 2 && Shape::CONV && NW == 8 && RS == 1 && QT == 1 && R == MF && (CPR / CPS) % 2 == 0 && (KS * CPS * 16) % 256 == 0), "K parts");
    // RS > 1 ("row split"): RS consecutive waves share one query group and take every RS-th row tile of the stage each,
    // so that few queries still spread their MFMAs over all four SIMDs (fp32 MFMAs bind long before HBM does)
    constexpr int RT = R / MF / RS;             // MFMA row tiles per stage and wave
    constexpr int STAGE = R * ROWB;             // bytes of V per stage
    constexpr int PPL = R * CPR / 64 / 4;       // LDS-DMA pieces (1 KiB) per staging wave and tile: waves 4-7 stage
    constexpr bool AUX0 = METRIC != 0;
    constexpr int NAUX = (AUX0 ? 1 : 0) + (HAS_BIAS ? 1 : 0);            // per-row aux values, staged by every B wave
    constexpr int QPW = MF * QT;                // queries per wave (QT query tiles share every A fragment)
    static_assert(R % (MF * RS) == 0 && 8 % RS == 0 && R <= 64 && (R * CPR) % 256 == 0 && ROWB % 256 == 0 && PPL + NAUX <= 31, "tile geometry");
    static_assert(NW == 8 || (NW == 4 && MODE == 1 && RS == 1), "four waves: the filter pass only");
    static_assert(RS == 1 || RS == 2, "MODE 2 publishes 8 granules per query and workgroup: 4 lanes x 2 values, or 2 x 4 lanes x 1");
    static_assert(MODE != 3 || (KSL && METRIC == 0 && !HAS_BIAS), "MODE 3 = raw partial sums of a K slice");
    static_assert(!KSL || (MODE != 2 && NW == 8), "K slices run in the multi-kernel pipeline");

    constexpr bool FILT = MODE == 1 || MODE == 2;   // the pass over all rows filters against per-query thresholds
    constexpr bool ONE = MODE == 2;             // the whole call in this launch
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* auxbuf = reinterpret_cast<float*>(smem + 3 * STAGE);                    // [3 stages][2][64]
    unsigned long long* cb = reinterpret_cast<unsigned long long*>(smem + 3 * STAGE + 3 * 2 * 64 * 4);
    unsigned short* cbq = reinterpret_cast<unsigned short*>(cb + HDB_MFMA_CB);
    unsigned int* ctl = reinterpret_cast<unsigned int*>(cbq + HDB_MFMA_CB);       // [0] count, [1..2] flush flags
    // MODE 2: per-query values every workgroup keeps for the exchange and the finish, BEHIND everything the final sort
    // overlays (mfma_batch_lds_bytes): [256] thresholds, [256] ||q||^2, [8] flags
    constexpr int XOFF = (3 * STAGE + 3 * 2 * 64 * 4 + HDB_MFMA_CB * 10 + 64) > (HDB_CAND_CAP * 16 + 2048 * 4 + 64)
                             ? (3 * STAGE + 3 * 2 * 64 * 4 + HDB_MFMA_CB * 10 + 64) : (HDB_CAND_CAP * 16 + 2048 * 4 + 64);
    float* xthr = reinterpret_cast<float*>(smem + XOFF);
    float* xss = xthr + HDB_BATCH_MAXQ;
    unsigned int* xflag = reinterpret_cast<unsigned int*>(xss + HDB_BATCH_MAXQ);   // [1] aborted (finish), [2] last to leave; [16 .. 80) owners' partial maxima

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    HDB_BSTAMP(0);
    const int rl = lane & (MF - 1);             // row of the A fragment == query of the B fragment
    const int h = lane / MF;                    // which 8-element k-chunk of the step (0..CPS-1)

    // ---- this wave's queries --------------------------------------------------------------------
    const int part = RS > 1 ? w % RS : 0;       // which row tiles of a stage this wave multiplies: part, part + RS, ...
    const int qw0 = a.q0 + blockIdx.y * ((NW / RS / KP) * QPW) + ((w % (NW / KP)) / RS) * QPW;
    const bool kp_upper = KP == 2 && w >= NW / 2;       // this wave multiplies the second half of the k-steps and has no epilogue
    const int ks0 = kp_upper ? KS : 0;                  // first k-step (of the row's CPR / CPS) of this wave
    const bool wave_active = qw0 < nq_end;
    bool q_ok[QT];
    int ql[QT];
    BVec Bq[QT][KS];
    float thr_l[QT], qinv_l[QT], qsq_l[QT];
    if constexpr (!ONE) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = qw0 + qt * MF + rl;
        q_ok[qt] = q < nq_end;
        ql[qt] = q - a.q0;
        const int qq = q_ok[qt] ? q : (nq_end - 1);
        const uint4* src = reinterpret_cast<const uint4*>(KSL ? q16 + (int64_t)qq * a.ks_dfull + a.ks_off / ES : q16 + (int64_t)qq * D) + RPF * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 v[RPF];
#pragma unroll
            for (int u = 0; u < RPF; ++u) {
                const bool past = KSL && a.ks_valid > 0 && (CPS * (s + ks0) + RPF * h + u) * 16 >= a.ks_valid;      // rows narrower than the geometry: nothing there
                v[u] = (past || !q_ok[qt]) ? make_uint4(0, 0, 0, 0) : src[CPS * (s + ks0) + u];
            }
            if constexpr (RPF == 1) Bq[qt][s] = *reinterpret_cast<Vec*>(&v[0]);
            else Bq[qt][s] = hdb_split3(HdbRaw8{*reinterpret_cast<f32x4*>(&v[0]), *reinterpret_cast<f32x4*>(&v[RPF - 1])});
        }
        thr_l[qt] = 0.f; qinv_l[qt] = 1.f; qsq_l[qt] = 0.f;
        if (q_ok[qt]) {
            if (MODE == 1) thr_l[qt] = a.thr[ql[qt]];
            // qscl = 2^-e: the fp16 queries were scaled by 2^e into [2^14, 2^15) (hdb_q16_scaled), undone here for free:
            // it rides on the per-query multiplier of dot / cosine and on the -2 of the euclidean expansion
            const float qs = qscl ? qscl[q] : 1.f;             // fp32 data: the queries are used as they are
            qinv_l[qt] = METRIC == 1 ? a.qinv[q] * qs : qs;
            if (METRIC == 2) qsq_l[qt] = qsq[q];
        }
    }
    } else {
        // ---- MODE 2: query preparation by the wave that multiplies with them (hdb_qprep_kernel, hdb_q16_scale) ----
        // The sum of squares is accumulated exactly as the prep kernel does it (lane e, e + 64, ... by fma, then the xor
        // butterfly), so 1/||q||, ||q||^2 and the NaN flag are bit-identical with the multi-kernel pipeline; the maximum
        // (the fp16 scale) does not depend on the order.  Lane (rl, h) keeps the values of query qt * MF + rl.
        const float* Qf = static_cast<const float*>(f.Qraw);
        const bool centre = f.centre != 0;          // pearson: subtract the query's mean first, exactly as hdb_qcentre_kernel does (same partial sums, same tree)
        float ss_l[QT], amax_l[QT], mean_l[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int q = qw0 + qt * MF + rl;
            q_ok[qt] = q < nq_end;
            ql[qt] = q - a.q0;
            ss_l[qt] = 0.f; amax_l[qt] = 0.f; mean_l[qt] = 0.f;
        }
        constexpr int QPL = (D + 63) / 64;
        constexpr int QB = QPW >= 32 ? 16 : 8;                // queries in flight per step
        if (wave_active) {
#pragma unroll 1
            for (int qi0 = 0; qi0 < QPW; qi0 += QB) {
                float xs[QB][QPL];
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    const int q = qw0 + qi0 + j;
                    const float* qv = Qf + (int64_t)(q < nq_end ? q : nq_end - 1) * D;
#pragma unroll
                    for (int u = 0; u < QPL; ++u) { const int e = lane + 64 * u; xs[j][u] = e < D ? qv[e] : 0.f; }
                }
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    float mean = 0.f;
                    if (centre) {
                        float sm = 0.f;
#pragma unroll
                        for (int u = 0; u < QPL; ++u) sm += xs[j][u];
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
                        mean = sm / (float)D;
#pragma unroll
                        for (int u = 0; u < QPL; ++u) xs[j][u] = lane + 64 * u < D ? xs[j][u] - mean : 0.f;
                    }
                    float ss = 0.f, am = 0.f;
#pragma unroll
                    for (int u = 0; u < QPL; ++u) { ss = fmaf(xs[j][u], xs[j][u], ss); am = fmaxf(am, fabsf(xs[j][u])); }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); am = fmaxf(am, __shfl_xor(am, o, 64)); }
                    const int qi = qi0 + j;
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        if (qi == qt * MF + rl) { ss_l[qt] = ss; amax_l[qt] = am; mean_l[qt] = mean; }
                }
            }
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int q = qw0 + qt * MF + rl;
            const int qq = q_ok[qt] ? q : (nq_end - 1);
            float scale = 1.f;
            if constexpr (ES == 2) scale = hdb_q16_scale(amax_l[qt]);
            if (wave_active) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if constexpr (ES == 2) {
                        const float4* src = reinterpret_cast<const float4*>(Qf + (int64_t)qq * D + (CPS * s + h) * 8);
                        float4 x0 = src[0], x1 = src[1];
                        if (centre) {
                            const float mq = mean_l[qt];
                            x0.x -= mq; x0.y -= mq; x0.z -= mq; x0.w -= mq; x1.x -= mq; x1.y -= mq; x1.z -= mq; x1.w -= mq;
                        }
                        Vec v;
                        v[0] = (_Float16)(x0.x * scale); v[1] = (_Float16)(x0.y * scale); v[2] = (_Float16)(x0.z * scale); v[3] = (_Float16)(x0.w * scale);
                        v[4] = (_Float16)(x1.x * scale); v[5] = (_Float16)(x1.y * scale); v[6] = (_Float16)(x1.z * scale); v[7] = (_Float16)(x1.w * scale);
                        if (!q_ok[qt]) v = Vec{0, 0, 0, 0, 0, 0, 0, 0};
                        Bq[qt][s] = v;
                    } else if constexpr (RPF == 2) {
                        const float4* src = reinterpret_cast<const float4*>(Qf + (int64_t)qq * D + (CPS * (s + ks0) + 2 * h) * 4);
                        float4 x0 = src[0], x1 = src[1];
                        if (centre) {
                            const float mq = mean_l[qt];
                            x0.x -= mq; x0.y -= mq; x0.z -= mq; x0.w -= mq; x1.x -= mq; x1.y -= mq; x1.z -= mq; x1.w -= mq;
                        }
                        HdbRaw8 v = {f32x4{x0.x, x0.y, x0.z, x0.w}, f32x4{x1.x, x1.y, x1.z, x1.w}};
                        if (!q_ok[qt]) v = HdbRaw8{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                        Bq[qt][s] = hdb_split3(v);
                    } else {
                        float4 x0 = *reinterpret_cast<const float4*>(Qf + (int64_t)qq * D + (CPS * s + h) * 4);
                        if (centre) { const float mq = mean_l[qt]; x0.x -= mq; x0.y -= mq; x0.z -= mq; x0.w -= mq; }
                        Vec v = {x0.x, x0.y, x0.z, x0.w};
                        if (!q_ok[qt]) v = Vec{0.f, 0.f, 0.f, 0.f};
                        Bq[qt][s] = v;
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if constexpr (ES == 2) Bq[qt][s] = Vec{0, 0, 0, 0, 0, 0, 0, 0};
                    else if constexpr (RPF == 2) Bq[qt][s] = HdbParts3{u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
                    else Bq[qt][s] = Vec{0.f, 0.f, 0.f, 0.f};
                }
            }
            thr_l[qt] = 0.f; qinv_l[qt] = 1.f; qsq_l[qt] = 0.f;
            if (q_ok[qt]) {
                const float ss = ss_l[qt];
                const float qs = 1.f / scale;                                      // a power of two: exact
                float qi = (ss == 0.f) ? 1.0f : 1.0f / sqrtf(ss);
                if (centre) { const float sd = sqrtf(ss / (float)D); qi = (sd == 0.f) ? __builtin_nanf("") : 1.0f / sd; }      // 1/sd_q, NaN for a constant query (:107-111)
                qinv_l[qt] = METRIC == 1 ? qi * qs : qs;
                if (METRIC == 2) qsq_l[qt] = ss;
                if (h == 0) xss[ql[qt]] = ss;                                       // for the finish (NaN flag, euclidean re-score)
            }
        }
        if (tid < 8) xflag[tid] = 0u;
    }
    if (tid < 4) ctl[tid] = 0;

    // ---- roles ----------------------------------------------------------------------------------------
    // Waves 0-3 ("A") and 4-7 ("B") are the two waves of each SIMD.  B runs its threshold epilogue one tile
    // late so that the two waves of a SIMD do not reach MFMA phase, epilogue and barrier in lock-step.
    const bool grpB = NW == 4 || w >= 4;          // the waves that stage (all four of a four-wave workgroup)
    const bool defer = KP == 2 ? false : NW == 4 ? w >= 2 : w >= 4;  // ... and the ones whose threshold epilogue runs one tile late
    HDB_BSTAMP(1);
    // B also stages every tile (see the note on staging roles at the top of this file); with up to 4*MF*QT queries in a
    // pass the B waves have no queries and do nothing else.

    // loop-invariant scalars, read once (keeps kernel-argument loads out of the tile loop)
    const char* const Vb = reinterpret_cast<const char*>(a.V);
    const int64_t n_rows = a.n;
    int64_t ntiles = ONE ? f.s_tiles : a.ntiles;                // tiles of the current pass (MODE 2: the sample first)
    int64_t tstride = ONE ? f.s_stride : a.tile_stride;         // 1 = dense pass, > 1 = strided row sample
    uint32_t* const tile_ctr = ONE ? (a.tile_ctr ? f.ctl + HDB_BATCH_CTL_TILE : nullptr) : a.tile_ctr;   // MODE 2: a.tile_ctr != nullptr = "hand tiles out dynamically"
    uint32_t* const gcnt = ONE ? f.ctl + HDB_BATCH_CTL_CNT : a.cnt;      // candidates appended per query ...
    constexpr int GCS = ONE ? 1 : HDB_CNT_STRIDE;                       // ... at this stride (the multi-kernel pipeline keeps one counter per cache line)
    const int64_t gstep = gridDim.x;

    // Stage tile number t (global tile index) into ring slot st: waves 4-7 issue PPL LDS-DMA pieces of 1 KiB each
    // (piece p = (w & 3) + 4 j; its source offset carries the XOR swizzle of the chunk), non-temporal (V is read once per
    // pass by exactly one CU), plus the per-row aux values.  Only the last tile of the matrix can be ragged.
    auto issue_aux = [&](int64_t row0, int st) {
        if ((AUX0 || HAS_BIAS) && grpB) {
            const int64_t last = n_rows - 1 - row0;
            const int64_t rr = lane <= last ? lane : last;
            if (AUX0) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(aux0g + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 0) * 64), 4, 0, 0);
            if (HAS_BIAS) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(a.bias + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 1) * 64), 4, 0, 0);
        }
    };
    auto issue_rows = [&](int64_t row0, int st) {
        const int64_t last = n_rows - 1 - row0;
        char* sdst = smem + st * STAGE;
        const int64_t pitch = KSL ? a.ks_pitch : (int64_t)ROWB;
        const char* tile_base = Vb + row0 * pitch + (KSL ? a.ks_off : 0);
        // (float32 tiles that become bf16 parts: EVERY wave stages an eighth of the tile -- pieces w, w + 8, ... -- and converts it, see
        //  convert_own; otherwise waves 4-7 stage a quarter each)
#pragma unroll
        for (int j = 0; j < (CONV ? PPL / 2 : PPL); ++j) {
            const int pc = CONV ? w + 8 * j : (w & 3) + 4 * j;
            const int slot = pc * 64 + lane;
            const int r = slot / CPR, cpos = slot - r * CPR;
            const int rr = r <= (int)last ? r : (int)last;
            int ch = cpos ^ (r & 15);                                                  // which 16 bytes of the row sit at this place of its LDS image
            if (KSL && a.ks_valid > 0 && ch * 16 >= a.ks_valid) ch = 0;                // past the end of a narrower row: its first chunk again (finite, times a zero query fragment)
            const unsigned int off = (unsigned int)(rr * (int)pitch + ch * 16);
            __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(tile_base + off), HDB_LDS_PTR(sdst + pc * 1024), 16, 0, 2);
        }
    };

    // Survivors of the filter go to a WAVE-PRIVATE segment of the LDS candidate list (HDB_MFMA_SEG entries per wave): the
    // slot of a survivor is the wave's running count plus its rank among the lanes that hit in the same step (one ballot,
    // no LDS atomic and no wait on its return in the filter's slow path), and a wave empties its own segment into the
    // global per-query lists whenever it is three quarters full -- no workgroup barrier, no flush decision to agree on.
    // (Survivors cost the batched passes 5 % with the shared list: 1.77 -> 1.68 ms at 256 queries with the append knocked out.)
    int wcnt = 0;                                    // wave-uniform: entries in this wave's segment
    const unsigned int seg_cb = (unsigned int)(uintptr_t)HDB_LDS_PTR(cb) + (unsigned int)w * HDB_MFMA_SEG * 8u;
    const unsigned int seg_cbq = (unsigned int)(uintptr_t)HDB_LDS_PTR(cbq) + (unsigned int)w * HDB_MFMA_SEG * 2u;
    auto flush = [&]() {                             // this wave's segment -> a.cand, then empty
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // One atomic per DISTINCT query among the 64 entries of a step (its first lane adds the group's size and hands the
        // base to the others): an atomic per entry put 64 returning atomics on 8-32 addresses into one wave-instruction, which
        // the memory side serialises per address -- 10-38 us per flush while the other workgroups still stream
        // (profiles/r3_batch1_timeline.txt: N=1.25M, 8 queries, pass done -> arrived).
        for (int e0 = 0; e0 < wcnt; e0 += 64) {
            const int e = e0 + lane;
            const bool live = e < wcnt;
            unsigned long long ent = 0ull; unsigned int qe = 0xFFFFu;
            if (live) asm volatile("ds_read_b64 %0, %2\n\tds_read_u16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                                   : "=&v"(ent), "=&v"(qe) : "v"(seg_cb + (unsigned int)e * 8u), "v"(seg_cbq + (unsigned int)e * 2u) : "memory");
            unsigned long long todo = __ballot(live);
            int leader = lane; unsigned int rank = 0u, gsize = 0u;
            while (todo) {
                const int first = (int)__ffsll((long long)todo) - 1;
                const unsigned int fq = (unsigned int)__builtin_amdgcn_readlane((int)qe, first);
                const unsigned long long grp = __ballot(live && qe == fq);
                if (live && qe == fq) {
                    leader = first;
                    rank = (unsigned int)__popcll(grp & ((1ull << lane) - 1ull));
                    gsize = (unsigned int)__popcll(grp);
                }
                todo &= ~grp;
            }
            unsigned int base = 0u;
            if (live && leader == lane) base = atomicAdd(&gcnt[qe * GCS], gsize);
            base = (unsigned int)__shfl((int)base, leader, 64);
            const unsigned int pos = base + rank;
            if (live && pos < a.cap) a.cand[(int64_t)qe * a.cap + pos] = ent;
        }
        wcnt = 0;
    };

    // ---- tile sequence ---------------------------------------------------------------------------------
    // Static: workgroup b takes tiles b, b+G, ...  Dynamic (filter pass of up to 4*MF*QT queries, one query block, a zeroed
    // counter in a.tile_ctr, >= 32 tiles per workgroup): the first chunk of CH tiles is fixed, later chunks come from the
    // counter (CH tiles per request, CH/2 near the end) -- CUs differ in streaming speed (hdb_mfma_fused.h) and the slowest
    // workgroup of a static split finishes 3-5 % of the pass after the median one.  Wave 3 owns the counter: it is the last
    // of the multiplying waves to get queries, so up to 3*MF*QT queries it has nothing else to do, and beyond that it
    // multiplies for a quarter of a round; waves 4-7, which stage, never wait for the counter.  The request goes out LOOK
    // rounds (~3 us of streaming, the counter answers in ~1.5 us) before the chunk is needed; the answer is handed over
    // through LDS.
    const bool heavy = NW == 4 || KP == 2 || (nq_end - (a.q0 + (int)blockIdx.y * ((8 / RS) * QPW))) > (4 / RS) * QPW;    // all waves multiply
    const int64_t G = gstep, bidx = blockIdx.x;
    // Measured (10 M rows, 8-64 queries, static -> dynamic): d=768 2.29 -> 2.18 ms, d=1536 (2.5 M rows) 1.149 -> 1.109,
    // d=512 (5 M) 0.775 -> 0.764, d=384 1.120 -> 1.106; but d=128 401 -> 438 us, d=256 (5 M) 394 -> 404, d=384 at 2.5 M rows
    // 298 -> 303: short rows (rounds shorter than the hand-over) and short passes (the end of a pass is decided in chunks)
    // lose, so dynamic hand-out needs rows of >= 768 bytes and >= 16 MiB of V per workgroup.
    constexpr int LOOK = STAGE >= 48 * 1024 ? 2 : STAGE >= 32 * 1024 ? 3 : STAGE >= 24 * 1024 ? 4 : STAGE >= 16 * 1024 ? 6 : 8;
    constexpr int64_t CH_ = 2 * LOOK;
    auto dyn_ok = [&]() { return FILT && tile_ctr != nullptr && tstride == 1 && gridDim.y == 1 && (!heavy || a.dyn_heavy) && ROWB >= 768 &&
                                 ntiles * STAGE >= G * a.dyn_min_bytes && ntiles >= 4 * CH_ * G && !(HDB_MFMA_KNOCKOUT & 2); };
    bool dyn = !ONE && dyn_ok();                     // MODE 2: decided when the filter pass begins (the sample is split statically)
    const int64_t CH = 2 * LOOK, dyn0 = G * CH;
    unsigned int* dq = ctl + 4;                      // [2] {first tile - dyn0, length} handed over by wave 3
    const unsigned int dq_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(dq);
    int64_t gp = 0, cidx = 0, coff = 0, cbase = 0, clen = CH, seen = 0;
    unsigned int req_got = 0u, req_want = 0u, req_slot = 0u; bool req_pending = false;     // wave 3: a chunk request in flight
    auto gen = [&](int64_t& t, int64_t& row0, bool& valid) {     // -> tile and row0 of sequence position gp (and whether it exists)
        if (!dyn) {
            t = bidx + gp * G;
            valid = bidx + gp * G < ntiles;
        } else {
            if (coff == 0) {
                if (cidx == 0) { cbase = bidx * CH; clen = CH; }
                else {
                    unsigned long long pr;
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pr) : "v"(dq_addr + (unsigned int)(cidx & 1) * 8u) : "memory");
                    cbase = dyn0 + (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)pr);
                    clen = (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(pr >> 32));
                }
            }
            // Wave 3 requests the next chunk LOOK rounds before it is needed and hands the answer over one round later (LOOK >= 2:
            // a barrier still lies between that LDS write and its readers), so that it never waits for the atomic: with all
            // eight waves multiplying a wait here would hold every round's barrier up.
            if (w == 3) {
                if (req_pending) {
                    const unsigned int got = (unsigned int)__builtin_amdgcn_readfirstlane((int)req_got);
                    seen = (int64_t)got + req_want;
                    const unsigned long long pr = ((unsigned long long)req_want << 32) | got;
                    if (lane == 0) asm volatile("ds_write_b64 %0, %1" :: "v"(dq_addr + req_slot * 8u), "v"(pr) : "memory");
                    req_pending = false;
                }
                if (coff == clen - LOOK) {
                    req_want = ntiles - dyn0 - seen > (CH + LOOK) * G ? (unsigned int)CH : (unsigned int)LOOK;
                    req_got = 0u;
                    if (lane == 0) req_got = __hip_atomic_fetch_add(tile_ctr, req_want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    req_slot = (unsigned int)((cidx + 1) & 1);
                    req_pending = true;
                }
            }
            t = cbase + coff;
            valid = t < ntiles;
            if (++coff == clen) { coff = 0; ++cidx; }
        }
        row0 = hdb_tile_index(t, tstride) * R;
        ++gp;
    };
    auto issue = [&](int64_t row0, int st) {
        if ((HDB_MFMA_KNOCKOUT & 2) && gp > 3) return;           // knock-out: the ring keeps its first three tiles
        if (grpB || CONV) issue_rows(row0, st);
        if (grpB) issue_aux(row0, st);
    };
    // CONV (float32 rows as bf16 parts, MfmaShape<16, hdb_f32s>): every wave stages an eighth of each tile and turns the pieces IT
    // staged into [a0 x 8][a1 x 8] groups, in place, as soon as its own vmcnt says they have landed -- one tile ahead of the
    // multiplication, so the round's one barrier still separates writer and readers.  (First version: waves 4-7 staged and converted
    // a quarter each -- their round was stage 1 090 + convert 2 730 clocks against 2 270 of multiplying in waves 0-3, which waited
    // 1 400 at the barrier: profiles/r4_f32_round_sections.txt.)  32-byte slot sl of a stage (row r = sl / (CPR/2)) holds the chunks 2g and
    // 2g+1 of its row, the odd one first where the row's swizzle (r & 15) is odd.  Two 1-KiB pieces = 64 slots per step.
    auto convert_own = [&](int st) __attribute__((always_inline)) {
        if constexpr (CONV && !(HDB_MFMA_KNOCKOUT & 16)) {      // (knock-out 16: the tiles stay float32 bit patterns -- timing only)
            static_assert(NW == 8 && PPL % 4 == 0, "conversion shares");
            const unsigned int cbase = (unsigned int)(uintptr_t)HDB_LDS_PTR(smem) + (unsigned int)(st * STAGE);
            constexpr int NS = PPL / 4;                         // 32-byte slots per lane: two 1-KiB pieces = 64 slots per step
            unsigned int alo[NS];
            f32x4 clo[NS], chi[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const unsigned int pc = (unsigned int)(w + 8 * (2 * u + (lane >> 5)));
                const unsigned int sl = pc * 32u + (unsigned int)(lane & 31);
                // (the first k-chunk of every slot first: 16 consecutive lanes touch half of the LDS banks twice.  Alternating the order every
                //  8 slots -- with the query parts and the MFMA operands of those k-groups swapped to match -- was built and measured: no
                //  change, the conversion waits for its turn at the LDS behind the fragment reads, not for bandwidth.)
                const unsigned int odd = (sl / (CPR / 2)) & 1u;
                alo[u] = cbase + sl * 32u + odd * 16u;
                asm volatile("ds_read_b128 %0, %1" : "=v"(clo[u]) : "v"(alo[u]) : "memory");
                asm volatile("ds_read_b128 %0, %1" : "=v"(chi[u]) : "v"(alo[u] ^ 16u) : "memory");
            }
            static_assert(NS == 2 || NS == 3, "conversion shares");        // (the outputs of all the reads hang on this wait: nothing of them moves above it)
            if constexpr (NS == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(clo[0]), "+v"(chi[0]), "+v"(clo[1]), "+v"(chi[1]) :: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(clo[0]), "+v"(chi[0]), "+v"(clo[1]), "+v"(chi[1]), "+v"(clo[NS - 1]), "+v"(chi[NS - 1]) :: "memory");
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                u32x4 a0, a1;
                hdb_round2(clo[u], chi[u], a0, a1);
                asm volatile("ds_write_b128 %0, %1" :: "v"(alo[u]), "v"(a0) : "memory");
                asm volatile("ds_write_b128 %0, %1" :: "v"(alo[u] ^ 16u), "v"(a1) : "memory");
            }
        }
    };
    int64_t tA, tB, tC = 0, rA, rB, rC = 0; bool vA, vB, vC = false;       // tile / first row / existence of sequence positions i, i+1, i+2
    int st_cur = 0;                                  // ring slot of sequence position i
    bool had_tiles = false;
    // Start a pass: the first two tiles of its sequence go into slots st_cur and st_cur + 1 (a second pass may start
    // without a barrier: the slot the previous pass used last is the third one, and every wave is past the barrier of the
    // round before that).
    auto begin_pass = [&]() __attribute__((always_inline)) {
        gp = 0; cidx = 0; coff = 0; cbase = 0; clen = CH; seen = 0; req_pending = false;
        gen(tA, rA, vA);
        gen(tB, rB, vB);
        had_tiles = vA;
        if (vA) issue(rA, st_cur);
        if (vB) issue(rB, st_cur == 2 ? 0 : st_cur + 1);
        if constexpr (CONV) {
            if (vA) {                                        // the first tile of the pass: converted before the first barrier
                if (!vB) hdb_wait_vmcnt<0>(); else if (grpB) hdb_wait_vmcnt<PPL / 2 + NAUX>(); else hdb_wait_vmcnt<PPL / 2>();
                convert_own(st_cur);
            }
        }
    };
    begin_pass();

    const unsigned int smem_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(smem);
    // per-lane LDS read address: row rl of a row tile; chunk (CPS*s + h) ^ rx of k-step s is at byte
    // ((16*CPS*s) ^ hx) of the row image, hx = (h ^ rx) << 4  (CPS*s and h occupy disjoint bits)
    const unsigned int rd_base = (unsigned int)((rl + part * MF) * CPR * 16) + (unsigned int)(ks0 * CPS * 16);     // (a multiple of 256: above the swizzle bits)
    const unsigned int hx = (unsigned int)(((RPF * h) ^ (rl & 15)) << 4);      // (RPF == 2: the lane's second chunk is at this address ^ 16)
    // first of the 4 consecutive tile rows this lane's accumulator group g holds
    auto grp_row = [&](int rt, int g) { const int rg = part + RS * rt; return MF == 32 ? rg * 32 + 8 * g + 4 * h : rg * 16 + 4 * h; };

    // threshold in the domain the epilogue compares in (see below); +inf for padding lanes
    float thr_cmp[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        thr_cmp[qt] = INFINITY;
        if (MODE == 1 && q_ok[qt]) {
            if (METRIC != 2 && !HAS_BIAS) { const float tc = thr_l[qt] / qinv_l[qt]; thr_cmp[qt] = tc - fabsf(tc) * 1e-6f; }
            else thr_cmp[qt] = thr_l[qt];
        }
    }
    // MODE 2, phase A: the two largest comparable values this lane has seen, per query tile (NaN counts as -inf)
    float top0[QT], top1[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { top0[qt] = -INFINITY; top1[qt] = -INFINITY; }
    auto sample_update = [&](const Acc (&tv)[QT][RT]) __attribute__((always_inline)) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 4 * NGRP; ++e) {
                    float v = tv[qt][rt][e];
                    v = (v != v) ? -INFINITY : v;
                    const float lo = fminf(top0[qt], v);
                    top0[qt] = fmaxf(top0[qt], v);
                    top1[qt] = fmaxf(top1[qt], lo);
                }
    };

    // Filter, second half: group maxima (v_max3) let the common no-hit case finish in ~25 VALU
    // instructions; survivors go to the workgroup's LDS list.  `tv` holds comparable values (below).
    auto filter1 = [&](const Acc (&tv)[RT], int64_t row0, const float thr_cmp, const float qinv_l, const int ql) {
        float gm[RT][NGRP];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < NGRP; ++g)
                gm[rt][g] = fmaxf(fmaxf(tv[rt][4 * g], tv[rt][4 * g + 1]), fmaxf(tv[rt][4 * g + 2], tv[rt][4 * g + 3]));
        float m = gm[0][0];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < NGRP; ++g) m = fmaxf(m, gm[rt][g]);
        // every branch below is wave-uniform (ballots): the running count stays a scalar
        if (__ballot((HDB_MFMA_KNOCKOUT & 1) ? (m == 1.2345e30f) : (m >= thr_cmp)) != 0ull) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int g = 0; g < NGRP; ++g) {
                    if (__ballot(gm[rt][g] >= thr_cmp) != 0ull) {
                        const int64_t rowg = row0 + grp_row(rt, g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x = tv[rt][4 * g + j];
                            const bool hit = x >= thr_cmp && rowg + j < n_rows && !(HAS_BIAS && x == -INFINITY);   // bias -inf = masked row
                            const unsigned long long act = __ballot(hit);
                            if (act != 0ull) {
                                const int pos = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)act, 0u));
                                if (hit) {
                                    const float sc = hdb_canon((METRIC != 2 && !HAS_BIAS) ? x * qinv_l : x);
                                    const unsigned long long ent = hdb_pack(sc, (uint32_t)(rowg + j));
                                    if (pos < HDB_MFMA_SEG) {
                                        // LDS ops in inline asm: hipcc would otherwise drain every in-flight LDS-DMA
                                        // (s_waitcnt vmcnt(0)) before touching LDS it cannot prove disjoint from the ring.
                                        asm volatile("ds_write_b64 %0, %1\n\tds_write_b16 %2, %3"
                                                     :: "v"(seg_cb + (unsigned int)pos * 8u), "v"(ent), "v"(seg_cbq + (unsigned int)pos * 2u), "v"((unsigned int)ql) : "memory");
                                    } else {         // cannot happen while flushes keep 64 slots free; kept as a safety net
                                        const unsigned int gpos = atomicAdd(&gcnt[ql * GCS], 1u);
                                        if (gpos < a.cap) a.cand[(int64_t)ql * a.cap + gpos] = ent;
                                    }
                                }
                                const int added = (int)__popcll(act);
                                wcnt = wcnt + added < HDB_MFMA_SEG ? wcnt + added : HDB_MFMA_SEG;
                                if (wcnt > HDB_MFMA_SEG - 64) flush();
                            }
                        }
                    }
                }
            }
        }
    };

    auto filter = [&](const Acc (&tv)[QT][RT], int64_t row0) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) filter1(tv[qt], row0, thr_cmp[qt], qinv_l[qt], ql[qt]);
    };

#if HDB_MFMA_CLOCK
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0) alone: keeps the loop's counted LDS waits as written
#endif
    Acc acc[QT][RT];
    int64_t row0_prev = 0;
#if HDB_ROUND_PROF
    unsigned long long rp_acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#endif
    // One pass over the current tile sequence.  ph = 0: MODE 2's sample pass (the epilogue feeds sample_update), otherwise the
    // pass the MODE names.
    auto run_pass = [&](auto ph) __attribute__((always_inline)) {
    constexpr bool SAMPLE = ONE && decltype(ph)::value == 0;
    auto epi = [&](const Acc (&tv)[QT][RT], int64_t row0) __attribute__((always_inline)) {
        if constexpr (SAMPLE) sample_update(tv); else filter(tv, row0);
    };
#if HDB_ROUND_PROF
    unsigned long long rp_t = __builtin_amdgcn_s_memtime();
#endif
    for (int64_t i = 0; vA; ++i) {
        if (!vB) hdb_wait_vmcnt<0>();
        else if (CONV) { if (grpB) hdb_wait_vmcnt<PPL / 2 + NAUX>(); else hdb_wait_vmcnt<PPL / 2>(); }
        else if (grpB) hdb_wait_vmcnt<PPL + NAUX>();         // all but the newest tile's pieces are in
        if (!(HDB_MFMA_KNOCKOUT & 4)) hdb_lds_barrier();     // tile i is in LDS; everyone is done with tile i-1; dq hand-over
        HDB_RP(0);
        // Stage the whole next-but-one tile right after the barrier, into the buffer tile i-1 used.
        const int st_next2 = st_cur == 0 ? 2 : st_cur - 1;
        if (vB) gen(tC, rC, vC); else vC = false;
        // CONV, one barrier per round: the two waves of a SIMD take the round's two halves in opposite order -- waves 4-7 stage and
        // convert first and multiply afterwards, waves 0-3 multiply first -- so that one wave's MFMAs run under the other's VALU / LDS
        // work instead of both queueing for the same unit (profiles/r4_f32_round_sections.txt)
        constexpr bool SWAP = CONV && KP == 1;
        auto stage_and_convert = [&]() __attribute__((always_inline)) {
            if (vC) issue(rC, st_next2);
            if constexpr (CONV) {
                if (vB) {                                    // tile i+1 (issued a round ago): this wave's pieces are in -> bf16 parts
#if HDB_ROUND_PROF
                    { const unsigned long long rp_now = __builtin_amdgcn_s_memtime(); rp_acc[grpB ? 1 : 5] += rp_now - rp_t; rp_t = rp_now; }      // [1] / [5]: staging alone
#endif
                    if (!vC) hdb_wait_vmcnt<0>(); else if (grpB) hdb_wait_vmcnt<PPL / 2 + NAUX>(); else hdb_wait_vmcnt<PPL / 2>();
#if HDB_ROUND_PROF
                    { const unsigned long long rp_now = __builtin_amdgcn_s_memtime(); rp_acc[7] += rp_now - rp_t; rp_t = rp_now; }                    // [7]: waiting for the pieces of tile i+1
#endif
                    convert_own(st_cur == 2 ? 0 : st_cur + 1);
#if HDB_ROUND_PROF
                    { const unsigned long long rp_now = __builtin_amdgcn_s_memtime(); rp_acc[grpB ? 5 : 1] += rp_now - rp_t; rp_t = rp_now; }      // conversion alone (booked on the other slot)
#endif
                }
            }
        };
        if constexpr (SWAP) { if (grpB) stage_and_convert(); }
        else if (vC) issue(rC, st_next2);
        HDB_RP(1);

        if (wave_active) {
            const int64_t row0 = rA;
            if (FILT && defer && i > 0) epi(acc, row0_prev);               // deferred epilogue of tile i-1
            // A fragments: LDS reads issued two k-steps ahead of the MFMAs that consume them.  The reads
            // and their counted waits are inline asm so that hipcc cannot sink a read next to its use
            // (it otherwise emits read, lgkmcnt(0), MFMA per step and exposes the LDS latency every step).
            // lgkmcnt(n*RT) = "all but the n*RT newest LDS ops are back" = the oldest pending step's fragments;
            // stray scalar loads can only make that wait longer, never shorter.
            const unsigned int sb_addr = smem_addr + (unsigned int)(st_cur * STAGE) + rd_base;
            constexpr int PF = (RPF == 2 && RT >= 2) ? 1 : (QT == 2 || RPF == 2) ? 2 : 3;  // k-steps of LDS prefetch (PF+1 fragment sets; two query tiles, two chunks per lane: registers)
            Vec abuf[PF + 1][RT];
            auto fetch = [&](int s, Vec (&dst)[RT]) {
                const unsigned int ad = sb_addr + ((unsigned int)(16 * CPS * s) ^ hx);
                if constexpr (RPF == 2) {
                    const unsigned int ad1 = ad ^ 16u;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0].lo) : "v"(ad));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0].hi) : "v"(ad1));
                    if constexpr (RT > 1) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1].lo) : "v"(ad), "i"(RS * MF * CPR * 16));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1].hi) : "v"(ad1), "i"(RS * MF * CPR * 16));
                    }
                    if constexpr (RT > 2) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[2].lo) : "v"(ad), "i"(2 * RS * MF * CPR * 16));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[2].hi) : "v"(ad1), "i"(2 * RS * MF * CPR * 16));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[RT - 1].lo) : "v"(ad), "i"(3 * RS * MF * CPR * 16));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[RT - 1].hi) : "v"(ad1), "i"(3 * RS * MF * CPR * 16));
                    }
                } else {
                asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(ad));
                if constexpr (RT > 1) { if ((HDB_MFMA_KNOCKOUT & 8) && RT == 4) dst[1] = dst[0]; else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(ad), "i"(RS * MF * CPR * 16)); }
                if constexpr (RT > 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[2]) : "v"(ad), "i"(2 * RS * MF * CPR * 16));
                if constexpr (RT > 3) { if (HDB_MFMA_KNOCKOUT & 8) dst[3] = dst[2]; else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[3]) : "v"(ad), "i"(3 * RS * MF * CPR * 16)); }
                }
            };
            // wait until at most `pend` k-steps of fragment reads are outstanding: lgkmcnt(pend*RT)
            auto wait_frag = [&](int pend, Vec (&f)[RT]) {
                static_assert(RT == 1 || RT == 2 || RT == 4, "RT");
#define HDB_WAITF(N)                                                                                               \
                do {                                                                                               \
                    if constexpr (RPF == 2) {                                                                      \
                        if constexpr (RT == 1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0].lo), "+v"(f[0].hi)); \
                        else if constexpr (RT == 2) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[RT - 1].lo), "+v"(f[RT - 1].hi)); \
                        else asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[RT - 2].lo), "+v"(f[RT - 2].hi), "+v"(f[RT - 1].lo), "+v"(f[RT - 1].hi)); \
                    } else                                                                                         \
                    if constexpr (RT == 1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]));                  \
                    else if constexpr (RT == 2) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]), "+v"(f[1])); \
                    else asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); \
                } while (0)
                const int cnt = pend * RPF * (((HDB_MFMA_KNOCKOUT & 8) && RT == 4) ? 2 : RT);
                if (cnt >= 12) HDB_WAITF(12); else if (cnt == 8) HDB_WAITF(8); else if (cnt == 6) HDB_WAITF(6);
                else if (cnt == 4) HDB_WAITF(4); else if (cnt == 3) HDB_WAITF(3); else if (cnt == 2) HDB_WAITF(2);
                else if (cnt == 1) HDB_WAITF(1); else HDB_WAITF(0);
#undef HDB_WAITF
            };
#pragma unroll
            for (int s = 0; s < PF && s < KS; ++s) fetch(s, abuf[s % (PF + 1)]);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int e = 0; e < 4 * NGRP; ++e) acc[qt][rt][e] = 0.f;
            if constexpr (KSL) {
                if (a.ks_partial_in && !kp_upper) {  // the sums of the K slices before this one (same layout as MODE 0's scores)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int g = 0; g < NGRP; ++g) {
                                const int rl0 = grp_row(rt, g);
                                const float* src = a.ks_partial_in + (int64_t)(q_ok[qt] ? ql[qt] : 0) * a.ks_ld + (tA * R + rl0);
                                if (q_ok[qt] && row0 + rl0 + 3 < n_rows) {
                                    const float4 pv = *reinterpret_cast<const float4*>(src);
                                    acc[qt][rt][4 * g] = pv.x; acc[qt][rt][4 * g + 1] = pv.y; acc[qt][rt][4 * g + 2] = pv.z; acc[qt][rt][4 * g + 3] = pv.w;
                                } else if (q_ok[qt]) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) if (row0 + rl0 + j < n_rows) acc[qt][rt][4 * g + j] = src[j];
                                }
                            }
                }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + PF < KS) fetch(s + PF, abuf[(s + PF) % (PF + 1)]);
                const int pend = (KS - 1 - s) < PF ? (KS - 1 - s) : PF;       // steps still in flight behind step s
                wait_frag(pend, abuf[s % (PF + 1)]);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) acc[qt][rt] = Shape::mma(abuf[s % (PF + 1)][rt], Bq[qt][s], acc[qt][rt]);
            }
            if constexpr (KP == 2) {                 // the second half of K: hand the partial sums over (this wave's own list segment)
                // (s_nop: the hazard recognizer does not see an inline-asm LDS store as a reader of the matrix pipe's result registers;
                //  up to 19 wait states between the last MFMA and a store of its accumulator, ISA guide 4.5)
                if (kp_upper) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3\n\tds_write_b128 %0, %1" :: "v"(seg_cb + (unsigned int)lane * 16u), "v"(acc[0][0]) : "memory");
            }
        }
        HDB_RP(2);
        if constexpr (KP == 2) hdb_lds_barrier();
        HDB_RP(3);
        if (wave_active && !kp_upper) {
            const int64_t row0 = rA;
            if constexpr (KP == 2) {
                f32x4 pv;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pv) : "v"(seg_cb + (unsigned int)(4 * HDB_MFMA_SEG * 8) + (unsigned int)lane * 16u) : "memory");
                acc[0][0] += pv;
            }

            // ---- epilogue, first half: turn the dot products into the values that are stored (MODE 0) or
            // compared (MODE 1), in place.  Filter mode compares the score itself, except dot / cosine without
            // bias: the raw dot (dot/||v||) against thr divided by the per-query multiplier.
            const float* ax0 = auxbuf + (st_cur * 2 + 0) * 64;
            const float* ax1 = auxbuf + (st_cur * 2 + 1) * 64;
            if (MODE != 3 && (METRIC != 0 || HAS_BIAS || MODE == 0)) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int g = 0; g < NGRP; ++g) {
                        const int rl0 = grp_row(rt, g);
                        float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (AUX0) av = *reinterpret_cast<const float4*>(ax0 + rl0);
                        if (HAS_BIAS) bv = *reinterpret_cast<const float4*>(ax1 + rl0);
                        const float aj[4] = {av.x, av.y, av.z, av.w};
                        const float bj[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float dot = acc[qt][rt][4 * g + j];
                                float x;
                                if (METRIC == 0) x = HAS_BIAS ? fmaf(dot, qinv_l[qt], bj[j]) : dot * qinv_l[qt];   // MODE 0 only without bias
                                else if (METRIC == 1) {
                                    if (FILT && !HAS_BIAS) x = dot * aj[j];
                                    else x = HAS_BIAS ? fmaf(dot * aj[j], qinv_l[qt], bj[j]) : dot * aj[j] * qinv_l[qt];
                                } else {
                                    const float d2 = fmaxf(fmaf(-2.f * qinv_l[qt], dot, aj[j] + qsq_l[qt]), 0.f);
                                    // v_sqrt_f32 / v_rcp_f32 (1 ulp each): the IEEE expansions of sqrtf and the division
                                    // cost ~25 VALU per score, on every row x query, for a result needed to 1e-3
                                    x = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_sqrtf(d2)) + (HAS_BIAS ? bj[j] : 0.f);
                                }
                                acc[qt][rt][4 * g + j] = x;
                            }
                        }
                    }
                }
            }
            if constexpr (MODE == 3) {                   // raw partial sums for the next K slice
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int g = 0; g < NGRP; ++g) {
                        const int rl0 = grp_row(rt, g);
                        const int64_t rowg = row0 + rl0;
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) {
                            if (q_ok[qt]) {
                                float* dst = a.ks_partial_out + (int64_t)ql[qt] * a.ks_ld + (tA * R + rl0);
                                if (rowg + 3 < n_rows) *reinterpret_cast<float4*>(dst) = make_float4(acc[qt][rt][4 * g], acc[qt][rt][4 * g + 1], acc[qt][rt][4 * g + 2], acc[qt][rt][4 * g + 3]);
                                else {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) if (rowg + j < n_rows) dst[j] = acc[qt][rt][4 * g + j];
                                }
                            }
                        }
                    }
            } else if (MODE == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int g = 0; g < NGRP; ++g) {
                        const int rl0 = grp_row(rt, g);
                        const int64_t rowg = row0 + rl0;
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) {
                            float sj[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) sj[j] = hdb_canon(acc[qt][rt][4 * g + j]);
                            if (q_ok[qt]) {
                                float* dst = a.scores + (int64_t)ql[qt] * a.ld + (tA * R + rl0);     // sample passes store compactly
                                if (rowg + 3 < n_rows) *reinterpret_cast<float4*>(dst) = make_float4(sj[0], sj[1], sj[2], sj[3]);
                                else {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) if (rowg + j < n_rows) dst[j] = sj[j];
                                }
                            }
                        }
                    }
                }
            } else {
                if (!defer) epi(acc, row0);
                else row0_prev = row0;
            }
        }
        HDB_RP(4);
        if constexpr (SWAP) { if (!grpB) stage_and_convert(); }
        else if constexpr (CONV) {
            if (vB) {                                        // tile i+1 (issued a round ago): this wave's pieces are in -> bf16 parts
                if (!vC) hdb_wait_vmcnt<0>(); else if (grpB) hdb_wait_vmcnt<PPL / 2 + NAUX>(); else hdb_wait_vmcnt<PPL / 2>();
                convert_own(st_cur == 2 ? 0 : st_cur + 1);
            }
        }
        HDB_RP(5);
#if HDB_ROUND_PROF
        rp_acc[6] += 1ull;
#endif
        st_cur = st_cur == 2 ? 0 : st_cur + 1;
        tA = tB; rA = rB; vA = vB; tB = tC; rB = rC; vB = vC;
    }
    if (FILT && wave_active && defer && had_tiles) epi(acc, row0_prev);    // the deferred epilogue of the last tile
    };
    if constexpr (!ONE) {
        run_pass(HdbIC<1>());
#if HDB_MFMA_CLOCK
        {
            const unsigned long long clk_c1 = __builtin_amdgcn_s_memtime(), clk_r1 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (tid == 0 && blockIdx.y == 0 && blockIdx.x < HDB_CLOCK_WGS) {
                unsigned long long* o = hdb_clock_buf + 4 * blockIdx.x;
                o[0] = clk_c0; o[1] = clk_c1; o[2] = clk_r0; o[3] = clk_r1;
            }
        }
#endif
#if HDB_ROUND_PROF
        if (lane == 0 && blockIdx.x < 256 && blockIdx.y == 0 && FILT) {
#pragma unroll
            for (int kx = 0; kx < 8; ++kx) hdb_round_prof[((int)blockIdx.x * 8 + w) * 8 + kx] = rp_acc[kx];
        }
#endif
        if (MODE == 1) flush();
    } else {
        // ================= MODE 2: sample pass, exchange, filter pass, finish =================
        const int nq_all = nq_end - a.q0;                     // queries of this launch (a.q0 == 0 here)
        hdb_gu64* const thrw = (hdb_gu64*)(reinterpret_cast<char*>(f.ctl) + HDB_BATCH_THRW_BYTE);
        auto expired = [&](unsigned long long t0) {
            return (unsigned long long)__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)f.timeout_ticks;
        };
        run_pass(HdbIC<0>());                                 // phase A
        HDB_BSTAMP(2);
        // the filter pass's first two tiles fly while the thresholds are agreed on
        ntiles = a.ntiles; tstride = 1; dyn = dyn_ok();
        begin_pass();
        // ---- publish: granules [query][workgroup][2 h], [.. + 1] = {epoch, key of this lane's largest, second largest}: ONE 16-byte
        // store per lane and query tile, the four lanes of a query fill its 64-byte line
        if (wave_active && !kp_upper) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (q_ok[qt]) {
                    const unsigned long long* dst = reinterpret_cast<const unsigned long long*>(f.ctl) + HDB_BATCH_GRAN_BYTE / 8 +
                                                    ((int64_t)ql[qt] * G + bidx) * 8 + (MF == 16 ? 2 * h : 4 * h);
                    if constexpr (RS == 2) {
                        // two waves share a query group (each multiplies every other row tile): eight lanes per query, ONE granule
                        // each -- the lane's largest value (both waves storing a pair to the same slots lost half of the sample:
                        // the threshold came out low enough to overflow a list about once in a hundred calls)
                        __hip_atomic_store((hdb_gu64*)(dst + part), ((unsigned long long)f.epoch << 32) | hdb_f2key(top0[qt]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        hdb_st128_sc1(dst, u32x4{hdb_f2key(top0[qt]), f.epoch, hdb_f2key(top1[qt]), f.epoch});
                        if (MF == 32) hdb_st128_sc1(dst + 2, u32x4{1u, f.epoch, 1u, f.epoch});    // two lane groups per query: four granules stay empty
                    }
                }
            }
        }
        const unsigned long long x_t0 = __builtin_amdgcn_s_memrealtime();
        HDB_BSTAMP(3);
        // ---- owners: query q belongs to workgroup q mod G.  The whole workgroup sweeps the G x 8 granules of an owned query
        // (one wave alone took ~10 us for the 16 KiB: four dependent batches of loads), every wave extracts the 8 largest of
        // its share, wave 0 the 8-th largest of those 64.
        {
            constexpr int M = 8;
            unsigned int* otop = xflag + 16;                  // [8 waves][M]
            for (int q = (int)bidx; q < nq_all; q += (int)G) {
                const int NG = (int)G * 8;
                bool gave_up = false;
#if HDB_BATCH_STAMPS
                unsigned long long attempts = 0ull;
#endif
                for (;;) {
                    bool ok = true;
#if HDB_BATCH_STAMPS
                    ++attempts;
#endif
                    uint32_t lmax[M];                         // per lane: the M largest of its granules, sorted descending
#pragma unroll
                    for (int r = 0; r < M; ++r) lmax[r] = 0u;
                    const char* srcb = reinterpret_cast<const char*>(f.ctl) + HDB_BATCH_GRAN_BYTE + (int64_t)q * G * 64;
                    for (int base = tid; base < NG / 2; base += 512 * 2) {       // pairs of granules, two 16-byte loads in flight per thread
                        const int i0 = base, i1 = base + 512 < NG / 2 ? base + 512 : base;
                        u32x4 xa, xb;
                        hdb_ld128x2_sc1(srcb + (int64_t)i0 * 16, srcb + (int64_t)i1 * 16, xa, xb);
                        auto take = [&](uint32_t key, uint32_t tag, bool dup) __attribute__((always_inline)) {
                            const bool tagged = tag == f.epoch;
                            ok &= tagged;
                            uint32_t v = (tagged && !dup) ? key : 0u;
#pragma unroll
                            for (int r = 0; r < M; ++r) { const uint32_t hi = max(lmax[r], v); v = min(lmax[r], v); lmax[r] = hi; }
                        };
                        take(xa.x, xa.y, false); take(xa.z, xa.w, false);
                        take(xb.x, xb.y, i1 == i0); take(xb.z, xb.w, i1 == i0);
                    }
#if HDB_BATCH_STAMPS
                    if (attempts == 1ull) HDB_BSTAMP(11);
#endif
                    // workgroup-wide votes through LDS words (xflag[3]: a granule is missing, xflag[4]: time is up)
                    if (tid == 0) { xflag[3] = 0u; xflag[4] = 0u; }
                    __syncthreads();
                    if (!ok) xflag[3] = 1u;
                    if (!ok && expired(x_t0)) xflag[4] = 1u;
                    __syncthreads();
                    if (xflag[3] == 0u) {
                        HDB_BSTAMP(13);
#if HDB_BATCH_STAMPS
                        if (tid == 0 && blockIdx.x < 1024) hdb_batch_stamps[16 * blockIdx.x + 12] = attempts;
#endif
#pragma unroll
                        for (int r = 0; r < M; ++r) {         // M rounds: the wave-wide maximum of the lane heads; its owner pops its head
                            const uint32_t v = hdb_wave_max_dpp(lmax[0]);
                            const unsigned long long who = __ballot(lmax[0] == v);
                            if (lane == (int)__ffsll((long long)who) - 1) {
#pragma unroll
                                for (int t = 0; t + 1 < M; ++t) lmax[t] = lmax[t + 1];
                                lmax[M - 1] = 0u;
                            }
                            if (lane == r) otop[w * M + r] = v;
                        }
                        break;
                    }
                    if (xflag[4] != 0u) { gave_up = true; break; }         // somebody never published
                    __syncthreads();                                      // (the words are reset after everybody has read them)
                    __builtin_amdgcn_s_sleep(4);
                }
                __syncthreads();
                HDB_BSTAMP(14);
                if (w == 0) {
                    uint32_t v = otop[lane], kth = 0u;
#pragma unroll
                    for (int r = 0; r < M; ++r) {
                        const uint32_t m = hdb_wave_max_dpp(v);
                        const unsigned long long who = __ballot(v == m);
                        if (lane == (int)__ffsll((long long)who) - 1) v = 0u;
                        kth = m;
                    }
                    if (kth <= 1u) kth = hdb_f2key(-INFINITY);          // fewer than M sample values exist: no threshold
                    if (gave_up) {                                       // nothing passes the filter; the host re-runs the call
                        kth = hdb_f2key(INFINITY);
                        if (lane == 0) atomicOr(f.ctl + HDB_BATCH_CTL_ABORT, 1u);
                    }
                    if (lane == 0) __hip_atomic_store(thrw + q, ((unsigned long long)f.epoch << 32) | kth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            }
        }
        HDB_BSTAMP(4);
        // ---- everybody: thread t fetches the threshold word of query t
        if (tid < nq_all) {
            unsigned long long v;
            for (;;) {
                v = __hip_atomic_load(thrw + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(v >> 32) == f.epoch) break;
                if (expired(x_t0)) {
                    v = hdb_f2key(INFINITY);
                    atomicOr(f.ctl + HDB_BATCH_CTL_ABORT, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            xthr[tid] = hdb_key2f((uint32_t)v);
        }
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) thr_cmp[qt] = q_ok[qt] ? xthr[ql[qt]] : INFINITY;      // already in the comparison domain
        HDB_BSTAMP(5);
#if HDB_MFMA_CLOCK
        const unsigned long long clk_c0b = __builtin_amdgcn_s_memtime(), clk_r0b = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
        run_pass(HdbIC<1>());                                 // phase B
#if HDB_MFMA_CLOCK
        {
            const unsigned long long clk_c1 = __builtin_amdgcn_s_memtime(), clk_r1 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (tid == 0 && blockIdx.x < HDB_CLOCK_WGS) {
                unsigned long long* o = hdb_clock_buf + 4 * blockIdx.x;
                o[0] = clk_c0b; o[1] = clk_c1; o[2] = clk_r0b; o[3] = clk_r1;
            }
        }
#endif
        HDB_BSTAMP(6);
#if HDB_ROUND_PROF
        if (lane == 0 && blockIdx.x < 256 && blockIdx.y == 0 && FILT) {
#pragma unroll
            for (int kx = 0; kx < 8; ++kx) hdb_round_prof[((int)blockIdx.x * 8 + w) * 8 + kx] = rp_acc[kx];
        }
#endif
        flush();
        // ---- finish: drain, release, arrive; wait for everybody; owners sort their queries
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(f.ctl + HDB_BATCH_CTL_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            HDB_BSTAMP(7);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool all_in = true;
            // (only the owners of a query wait for everybody -- workgroup b owns queries b, b + G, ...: with fewer queries than
            // workgroups the rest leave at once instead of polling the line the arrivals go to, see hdb_bits_fused.hip)
            while ((int64_t)bidx < (int64_t)nq_all && __hip_atomic_load(f.ctl + HDB_BATCH_CTL_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)G) {
                if (expired(t0)) { all_in = false; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (!all_in) atomicOr(f.ctl + HDB_BATCH_CTL_ABORT, 1u);
            const unsigned int ab = __hip_atomic_load(f.ctl + HDB_BATCH_CTL_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xflag[1] = (ab != 0u || !all_in) ? 1u : 0u;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const bool aborted = xflag[1] != 0u;
        HDB_BSTAMP(8);
        unsigned long long* fbuf = reinterpret_cast<unsigned long long*>(smem);     // the ring is free now
        for (int q = (int)bidx; q < nq_all; q += (int)G) {
            const uint32_t tot0 = __hip_atomic_load(gcnt + q * GCS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t tot = aborted ? 0u : tot0;
            const float ssq = xss[q];
            // euclidean scores come from ||v||^2 + ||q||^2 - 2 v.q, which cancels when v ~ q: candidates closer than 5 % of
            // ||q||^2 are re-scored from the stored row with the direct difference (hdb_rescore_euclid_kernel, reference :49)
            auto rescore = [&](unsigned long long* buf, uint32_t nc) {
                if constexpr (METRIC == 2) {
                    const float* qv = static_cast<const float*>(f.Qraw) + (int64_t)q * D;
                    const float close2 = 0.05f * ssq;
                    const E* Vr = static_cast<const E*>(a.V);
                    // every lane tests an entry of its own (64 per wave and step: nearly always none qualifies); the rare
                    // near-duplicate is then re-scored by the whole wave
                    for (uint32_t e0 = (uint32_t)w * 64u; e0 < nc; e0 += 8u * 64u) {
                        const uint32_t e = e0 + (uint32_t)lane;
                        const unsigned long long ent = e < nc ? buf[e] : 0ull;
                        const uint32_t row = 0xFFFFFFFFu - (uint32_t)(ent & 0xFFFFFFFFull);
                        const float s0 = hdb_key2f((uint32_t)(ent >> 32));
                        const float bb = (HAS_BIAS && e < nc) ? a.bias[row] : 0.f;
                        const float sim = s0 - bb;                             // 1 / (1 + dist) in (0, 1]; -inf for an excluded row
                        const float dist = 1.f / sim - 1.f;
                        unsigned long long todo = __ballot(e < nc && sim > 0.f && dist * dist < close2);
                        while (todo) {
                            const int src = (int)__ffsll((long long)todo) - 1;
                            todo &= todo - 1ull;
                            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)row, src);
                            float accd = 0.f;
                            for (int kx = lane; kx < D; kx += 64) { const float df = (float)Vr[(int64_t)r * D + kx] - qv[kx]; accd += df * df; }
#pragma unroll
                            for (int o = 32; o > 0; o >>= 1) accd += __shfl_xor(accd, o, 64);
                            if (lane == src) buf[e] = hdb_pack(hdb_canon(1.f / (1.f + sqrtf(accd)) + bb), row);
                        }
                    }
                    __syncthreads();
                }
            };
            if constexpr (METRIC == 2)
                hdb_finalize_body(fbuf, a.cand + (int64_t)q * a.cap, tot, q, a.cap, f.k, f.kk, f.row_base, f.idx_out, f.score_out, f.status,
                                  (ssq != ssq) ? 1 : 0, (RPF == 2 && ssq - ssq != 0.f) ? HDB_Q_UNDERFLOW : 0, nullptr, 1.f, rescore);
            else
                hdb_finalize_fast(fbuf, a.cand + (int64_t)q * a.cap, tot, q, a.cap, f.k, f.kk, f.row_base, f.idx_out, f.score_out, f.status,
                                  (ssq != ssq) ? 1 : 0, (RPF == 2 && ssq - ssq != 0.f) ? HDB_Q_UNDERFLOW : 0);      // parts of an infinite value cancel to NaN: exact re-run
            __syncthreads();
        }
        // ---- leave: the last workgroup out zeroes the control block for the next launch
        __syncthreads();
        HDB_BSTAMP(9);
        if (tid == 0) {
            const unsigned int left = __hip_atomic_fetch_add(f.ctl + HDB_BATCH_CTL_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xflag[2] = left == (unsigned int)G - 1u ? 1u : 0u;
        }
        __syncthreads();
        if (xflag[2]) {
            if (tid < nq_all) __hip_atomic_store(f.ctl + HDB_BATCH_CTL_CNT + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) {
                __hip_atomic_store(f.ctl + HDB_BATCH_CTL_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(f.ctl + HDB_BATCH_CTL_EXIT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(f.ctl + HDB_BATCH_CTL_TILE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(f.ctl + HDB_BATCH_CTL_ABORT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        HDB_BSTAMP(10);
    }
}

// ------------------------------------------------------------------------------------------------
static size_t mfma_lds_bytes(int stage_bytes) {
    return (size_t)3 * stage_bytes + 3 * 2 * 64 * 4 + (size_t)HDB_MFMA_CB * 8 + (size_t)HDB_MFMA_CB * 2 + 64;
}
// MODE 2: the scan's LDS or the final sort's (hdb_finalize_body overlays the ring), whichever is larger, plus the per-query
// values kept behind both (XOFF in the kernel)
static size_t mfma_batch_lds_bytes(int stage_bytes) {
    const size_t scan = mfma_lds_bytes(stage_bytes), fin = (size_t)HDB_CAND_CAP * 16 + 2048 * 4 + 64;
    return (scan > fin ? scan : fin) + 2 * HDB_BATCH_MAXQ * 4 + 512;
}

template <typename E, int MF, int QT, int D, int R, int RS, int MODE, int METRIC, bool HAS_BIAS, int KP = 1>
static int launch_one(const ScanArgs& a, const void* q16, const float* aux0, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st,
                      const BatchArgs* f) {
    auto kern = hdb_mfma_kernel<E, MF, QT, D, R, RS, MODE, METRIC, HAS_BIAS, 8, false, KP>;
    const size_t lds = MODE == 2 ? mfma_batch_lds_bytes(R * D * (int)sizeof(E)) : mfma_lds_bytes(R * D * (int)sizeof(E));
    static unsigned long long attr_done = 0;          // per instantiation, one bit per device
    hipError_t e = hdb_lds_attr_once(reinterpret_cast<const void*>(kern), (int)lds, &attr_done);
    if (e != hipSuccess) return (int)e;
    const dim3 grid(blocks, (nq_launch + (8 / RS / KP) * MF * QT - 1) / ((8 / RS / KP) * MF * QT));
    if (MODE == 2 && (grid.y != 1 || !f || a.q0 != 0)) return (int)hipErrorInvalidValue;
    BatchArgs fa = BatchArgs();
    if (f) fa = *f;
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a, (const E*)q16, aux0, qsq, qscl, a.q0 + nq_launch, fa);
    return (int)hipGetLastError();
}

// One K slice of a wide-row scan (hdb_mfma_ksplit.hip): mode 3 = raw partial sums out, 0 / 1 = the last slice with the metric's epilogue
template <typename E, int D, int R, int MODE, int METRIC, bool HAS_BIAS, int KP = 1>
static int launch_kslice_one(const ScanArgs& a, const void* q16, const float* aux0, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st) {
    auto kern = hdb_mfma_kernel<E, 16, 1, D, R, 1, MODE, METRIC, HAS_BIAS, 8, true, KP>;
    const size_t lds = mfma_lds_bytes(R * D * (int)sizeof(E));
    static unsigned long long attr_done = 0;
    hipError_t e = hdb_lds_attr_once(reinterpret_cast<const void*>(kern), (int)lds, &attr_done);
    if (e != hipSuccess) return (int)e;
    const dim3 grid(blocks, (nq_launch + 8 / KP * 16 - 1) / (8 / KP * 16));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a, (const E*)q16, aux0, qsq, qscl, a.q0 + nq_launch, BatchArgs());
    return (int)hipGetLastError();
}
template <typename E, int D, int R, int KP = 1>
static int launch_kslice(const ScanArgs& a, int mode, const void* q16, const float* sqnorm, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st) {
    if (mode == 3) return launch_kslice_one<E, D, R, 3, 0, false, KP>(a, q16, nullptr, qsq, qscl, nq_launch, blocks, st);
    const bool b = a.bias != nullptr;
#define HDB_KS_CASE(MODE_)                                                                                                                        \
    if (a.metric == HDB_DOT) return b ? launch_kslice_one<E, D, R, MODE_, 0, true, KP>(a, q16, nullptr, qsq, qscl, nq_launch, blocks, st)              \
                                      : launch_kslice_one<E, D, R, MODE_, 0, false, KP>(a, q16, nullptr, qsq, qscl, nq_launch, blocks, st);            \
    if (a.metric == HDB_COSINE) return b ? launch_kslice_one<E, D, R, MODE_, 1, true, KP>(a, q16, a.inv_norm, qsq, qscl, nq_launch, blocks, st)        \
                                         : launch_kslice_one<E, D, R, MODE_, 1, false, KP>(a, q16, a.inv_norm, qsq, qscl, nq_launch, blocks, st);      \
    if (a.metric == HDB_EUCLIDEAN) return b ? launch_kslice_one<E, D, R, MODE_, 2, true, KP>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st)         \
                                            : launch_kslice_one<E, D, R, MODE_, 2, false, KP>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st);
    if (mode == 0) { HDB_KS_CASE(0) } else { HDB_KS_CASE(1) }
#undef HDB_KS_CASE
    return (int)hipErrorNotSupported;
}

// the four-wave measurement variant of the filter pass (dot product, no bias)
template <typename E, int MF, int QT, int D, int R>
static int launch_four_waves(const ScanArgs& a, const void* q16, const float* qscl, int nq_launch, int blocks, hipStream_t st) {
    auto kern = hdb_mfma_kernel<E, MF, QT, D, R, 1, 1, 0, false, 4>;
    const size_t lds = mfma_lds_bytes(R * D * (int)sizeof(E));
    static unsigned long long attr_done = 0;
    hipError_t e = hdb_lds_attr_once(reinterpret_cast<const void*>(kern), (int)lds, &attr_done);
    if (e != hipSuccess) return (int)e;
    const dim3 grid(blocks, (nq_launch + 4 * MF * QT - 1) / (4 * MF * QT));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a, (const E*)q16, nullptr, nullptr, qscl, a.q0 + nq_launch, BatchArgs());
    return (int)hipGetLastError();
}

template <typename E, int MF, int QT, int D, int R, int RS, int MODE, int KP = 1>
static int launch_metric(const ScanArgs& a, const void* q16, const float* sqnorm, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st,
                         const BatchArgs* f) {
    const bool b = a.bias != nullptr;
    if (a.metric == HDB_DOT) return b ? launch_one<E, MF, QT, D, R, RS, MODE, 0, true, KP>(a, q16, nullptr, qsq, qscl, nq_launch, blocks, st, f)
                                      : launch_one<E, MF, QT, D, R, RS, MODE, 0, false, KP>(a, q16, nullptr, qsq, qscl, nq_launch, blocks, st, f);
    if (a.metric == HDB_COSINE) return b ? launch_one<E, MF, QT, D, R, RS, MODE, 1, true, KP>(a, q16, a.inv_norm, qsq, qscl, nq_launch, blocks, st, f)
                                         : launch_one<E, MF, QT, D, R, RS, MODE, 1, false, KP>(a, q16, a.inv_norm, qsq, qscl, nq_launch, blocks, st, f);
    if (a.metric == HDB_EUCLIDEAN) return b ? launch_one<E, MF, QT, D, R, RS, MODE, 2, true, KP>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f)
                                            : launch_one<E, MF, QT, D, R, RS, MODE, 2, false, KP>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
    return (int)hipErrorNotSupported;
}

// mode 2 (the whole call in one launch) takes BatchArgs; q16 / qsq / qscl are unused there (the kernel prepares the queries)
template <typename E, int MF, int QT, int D, int R, int RS = 1, int KP = 1>
static int launch_mode(const ScanArgs& a, int mode, const void* q16, const float* sqnorm, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st,
                       const BatchArgs* f = nullptr) {
    if (mode == 0) return launch_metric<E, MF, QT, D, R, RS, 0, KP>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, nullptr);
    if (mode == 2) return launch_metric<E, MF, QT, D, R, RS, 2, KP>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, f);
    return launch_metric<E, MF, QT, D, R, RS, 1, KP>(a, q16, sqnorm, qsq, qscl, nq_launch, blocks, st, nullptr);
}

