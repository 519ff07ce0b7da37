// hdb_mfma_fused.h -- ONE launch for a whole hdb_topk call of 1..4 queries on an fp16 matrix (dot / cosine):
// query preparation, strided row sample, threshold, the filter pass over all rows and the final top-k, which the
// multi-kernel pipeline (hdb_api.hip) does in five dependent launches (qprep -> sample scan -> sample threshold ->
// filter scan -> finalize, ~45 us of fixed cost of which ~17 us are queue gaps between the launches).  The reference
// does all of it in one Python call per query (hyperdb/ranking_algorithm.py:149-204).
//
// Roles of the eight waves (a wave that blocks on an exchange must not also be on the staging path):
//   wave 0 multiplies and filters; waves 1-2 are the selectors (two queries each); wave 3 requests tile chunks from the
//   global counter; waves 4-7 stage the tiles (LDS-DMA, pieces of 1 KiB each) and the per-row aux values.
// Structure (persistent, one 512-thread workgroup per CU, the LDS ring of hdb_mfma_kernel.h):
//   prologue  every workgroup converts the float32 queries itself (fp16 copies scaled by a power of two into LDS,
//             1/||q||, NaN flag): 4 x d elements, cheaper than a launch.
//   phase A   workgroup b multiplies sample tiles b, b+G, ... of the strided, jittered row sample (same plan as the
//             multi-kernel path); wave 0 holds the queries as MFMA B fragments, its scores of each tile go to a small
//             LDS buffer; waves 1..nq ("selectors", otherwise only staging) keep the m largest sample scores of their
//             query in registers (m rounds of wave-max extraction per tile).
//   exchange  each selector publishes its m values as 8-byte {epoch, value} granules (one sc1 store per lane; the
//             data is its own flag).  Workgroup 0 sweeps the granules of the workgroups that have sample tiles until
//             enough of them carry this call's epoch -- an all-gather without a grid barrier -- and stores the m-th
//             largest as one {final | epoch, key} word per query; every other workgroup polls that word (an LDS-DMA
//             load issued one round, looked at the next).  The m-th largest of ALL sample values is the exact sample
//             threshold; of a subset, a lower bound of it that is already safe to filter with.
//   parking   until its threshold arrives wave 0 keeps the scores of its filter tiles in LDS (up to 16 tiles) and
//             filters them in one go afterwards: the stream never stops for the exchange.
//   phase B   the filter pass over all rows (the only pass over V); wave 0 defers the epilogue of a tile by one tile.
//   finish    survivors go to the global candidate lists (atomics); each workgroup drains, releases and takes a
//             ticket; the LAST workgroup re-uses the ring LDS to pre-select, sort and write the k results (and the
//             status words) of every query, straight into the caller's (pinned host or device) buffers.
// Short matrices (round 4, FusedArgs::local): when all tiles of a workgroup fit its parking area (<= 16 tiles for 1-2 queries)
// there is no phase A and no exchange at all.  Wave 0 parks the scores of every tile; after the last tile wave q takes the
// local_m-th largest of the lane maxima of query q as the workgroup's own threshold b_w (at least local_m rows reach it),
// reserves its slots in the global list with ONE atomic and writes the rows at or above b_w straight from the parked scores.
// Any row a workgroup did not emit scores below that workgroup's b_w, so the union holds the global top-k as soon as every b_w
// lies strictly below the k-th best of the union: the last workgroup checks exactly that (the b_w travel as score-domain keys
// in the granule area) and reports UNDERFLOW otherwise -- the host then re-runs the call through the exact path, as for a
// failed sampled threshold.  Random rows: P(some workgroup holds local_m = 12 of the 99 best) ~ 1e-12 per call.
// Every spin is bounded by a wall-clock timeout (s_memrealtime): a workgroup that gives up publishes nothing harmful,
// filters with threshold +inf, and raises the abort word, which turns every status into HDB_Q_UNDERFLOW so that the
// host re-runs the call through the exact path (and resets the control block).  This only happens when the grid is
// not co-resident (another kernel holds CUs), never on an idle GPU: grid <= CU count, one workgroup fits per CU.
#pragma once
#include <hip/hip_runtime.h>

// Diagnostic build only (tools/stamps_fused.py; product: 0): wall-clock stamps (s_memrealtime, 100 MHz) of the phases
// of every workgroup, stored in a buffer of their own that nothing reads: [wg][8] = start, prologue done, published,
// threshold known, loop done, ticket taken, finalize done (last workgroup only), spare.
#define HDB_CLOCK_WGS_F 1024
#ifndef HDB_FUSED_STAMPS
#define HDB_FUSED_STAMPS 0
#endif
#if HDB_FUSED_STAMPS
static __device__ unsigned long long hdb_fused_stamps[16 * HDB_CLOCK_WGS_F];
#define HDB_STAMP(slot) do { if (lane == 0 && stamp_wave) hdb_fused_stamps[16 * blockIdx.x + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define HDB_FIN_STAMP(slot) do { if (threadIdx.x == 0) hdb_fused_stamps[16 * blockIdx.x + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HDB_STAMP(slot) do { } while (0)
#endif

#include "hdb_mfma_kernel.h"          // (includes hdb_finalize.h: the stamp macros above must come first)

#define HDB_FUSED_MAXQ 4            // queries per fused call
#define HDB_FUSED_M 8               // sample order statistic (k <= 128)
#define HDB_FUSED_GRAN_PER_WG 32    // HDB_FUSED_MAXQ * HDB_FUSED_M granules per workgroup
#define HDB_FUSED_MAX_WG 1024
// Control block (FusedArgs::ctl, zero when allocated, tagged by epoch afterwards), in bytes:
//   0   ticket, abort word, per-query candidate counts (uint32 [0..5])
//   128 tile counter (a line of its own: ~35-70 atomics per us)
//   256 threshold words of workgroup 0, {final << 31 | epoch, key} per query (a line of their own: a line that takes
//       atomics answers plain loads tens of us late)
//   512 HDB_FUSED_POLLS copies of those words, one 128-byte line each, [selector wave][2 queries] x 8 bytes: a poller
//       reads copy number `round`, so it never asks for a line its XCD's L2 may hold from before the words were written
//   HDB_FUSED_HDR_BYTES  the sample granules, [workgroup][HDB_FUSED_GRAN_PER_WG] x 8 bytes
#define HDB_FUSED_POLLS 40
#define HDB_FUSED_HDR_BYTES 8192
#define HDB_FUSED_PEND 8            // filter tiles whose scores can be parked in LDS while no threshold is known yet: this many for
                                    // the most queries a flavour takes, up to 16 for fewer (same bytes)


// E = _Float16: wave 0 multiplies on the matrix cores (fp16 copies of the queries as B fragments).
// E = float (the reference's default fp_precision, BASELINE config 2): wave 0 computes the float32 dot products in the
// VALU straight from the staged tile (16 lanes x 16 B per row and pass, four rows per 16-lane group, the DPP ownership
// butterfly of hdb_scan.hip): for 1-4 queries that is ~200 v_fma per 32-row tile, far below what an fp32 MFMA spends on
// its 16 query columns, and exact float32 arithmetic (1e-5 contract).
template <typename E, int VQ, int D, int R, int METRIC, bool HAS_BIAS>
__global__ __launch_bounds__(512) void hdb_mfma_fused_kernel(ScanArgs a, FusedArgs f, const float* __restrict__ aux0g) {
    constexpr bool VALU = sizeof(E) == 4;
    using Shape = MfmaShape<16, _Float16>;
    using Vec = typename Shape::Vec;
    using Acc = typename Shape::Acc;
    constexpr int MF = 16;
    constexpr int ROWB = D * (int)sizeof(E);
    constexpr int NJ = ROWB / 256;              // VALU: 16-byte chunks per lane and row (16 lanes cover a row)
    constexpr int NP = R / 16;                  // VALU: 16-row passes per tile ...
    constexpr bool TWOCW = NP >= 2;             // two computing waves (0 and 2) share the passes; 16-row tiles have one pass: wave 0 alone
    constexpr int NPW = TWOCW ? NP / 2 : 1;     // ... per computing wave
    // VQ (VALU flavour only): queries per call, 1 or 2 -- their chunks live in registers (VQ x D/16 of them), and a
    // single query must not pay for a second one
    constexpr int CPR = ROWB / 16;
    constexpr int CPS = Shape::CPS;
    constexpr int KS = CPR / CPS;
    constexpr int RT = R / MF;
    constexpr int STAGE = R * ROWB;
    // BLDS: beyond d = 768 the query fragments (d/8 registers) do not fit beside the selectors' state: they stay in LDS (up
    // to 2 queries, inside the candidate-list region, which then keeps 256 entries) and wave 0 reads one fragment per k-step
    constexpr bool BLDS = !VALU && D > 768;
    constexpr int CB = BLDS ? 256 : HDB_MFMA_CB;   // LDS candidate-list entries in use
    constexpr bool COSLIKE = METRIC == 1 || METRIC == 3;     // pearson = cosine on the centred query with the row scale 1/(sd_v d) and 1/sd_q (hdb_scan.hip)
    constexpr bool AUX0 = COSLIKE || (METRIC == 2 && sizeof(E) == 2);      // cosine: 1/||v||; pearson: 1/(sd_v d); euclidean on the matrix cores: ||v||^2
    constexpr int NGL = R * CPR / 256;          // pieces per staging wave and tile (waves 4-7)
    constexpr int NLOADA = NGL;
    constexpr int NLOADB = NGL + (AUX0 ? 1 : 0) + (HAS_BIAS ? 1 : 0);
    constexpr int M = HDB_FUSED_M;
    static_assert(R % MF == 0 && R <= 64 && (R * CPR) % 256 == 0 && ROWB % 256 == 0, "tile geometry");
    static_assert(!BLDS || (RT == 1 && 2 * D * 2 + 2048 <= HDB_MFMA_CB * 8), "LDS-resident fragments: 16-row tiles, two queries behind 2 KiB of list");
    static_assert(!VALU || (NJ * VQ <= 12 && (NP % 2 == 0 || NP == 1)), "float32 flavour: query chunks in registers, one or two computing waves");
    static_assert(METRIC >= 0 && METRIC <= 3, "dot / cosine / euclidean / pearson (matrix cores: ||v||^2 + ||q||^2 - 2 v.q with aux0 = ||v||^2; float32 VALU flavour: the direct sum of (v - q)^2 of hdb_scan.hip, nothing to re-score)");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* auxbuf = reinterpret_cast<float*>(smem + 3 * STAGE);                    // [3 stages][2][64]
    unsigned long long* cb = reinterpret_cast<unsigned long long*>(smem + 3 * STAGE + 3 * 2 * 64 * 4);
    unsigned short* cbq = reinterpret_cast<unsigned short*>(cb + HDB_MFMA_CB);
    unsigned int* ctl = reinterpret_cast<unsigned int*>(cbq + HDB_MFMA_CB);       // [0] count, [1..2] flush flags, [3] last-workgroup flag
    float* tsc = reinterpret_cast<float*>(ctl + 16);                               // [2][MAXQ][64] scores of the latest sample tiles
    float* qpar = tsc + 2 * HDB_FUSED_MAXQ * 64;                                   // [MAXQ] multiplier, [MAXQ] NaN flag, [MAXQ] threshold
    // [nq][D] queries in E: prologue scratch in ring slot 2 (not yet in use) -- or, BLDS, resident behind the first 2 KiB of the list
    char* qlds = BLDS ? reinterpret_cast<char*>(cb) + 2048 : smem + 2 * STAGE;
    // Parked scores (see "parking" in the tile loop): [HDB_FUSED_PEND] tiles x [queries] x [rows] floats.  The MFMA flavour
    // (4 queries x 64 rows = 1 KiB per tile) parks in the candidate list, which is empty until a threshold exists; the
    // float32 flavour (VQ queries x R rows) has an area of its own behind qpar.  tsc is free once the sample tiles are
    // merged and holds the first row of each parked tile.
    float* pbuf = VALU ? qpar + 16 : reinterpret_cast<float*>(cb);
    static_assert(HDB_FUSED_PEND * HDB_FUSED_MAXQ * 64 * 4 <= HDB_MFMA_CB * 8, "parked MFMA scores live in the candidate list");
    static_assert(!BLDS || 16 * 2 * R * 4 <= 2048, "parked scores of 16 tiles x 2 queries fit the first 2 KiB");
    constexpr bool PARK = VALU || (D <= 640 && !(METRIC == 2 && D > 512)) || BLDS;       // d = 768 (euclidean: d = 640 too) would spill with it: there the selector holds the round as before

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & (MF - 1);
    const int h = lane / MF;
    const int nq = f.nq;
    // parked tiles: the MFMA flavour packs nq x 64 floats per tile into the 8 KiB list, the float32 flavour VQ x R floats into its area
    const int pend_max = VALU ? 2 * HDB_FUSED_PEND / VQ : (nq <= 2 ? 16 : 32 / nq);
    const unsigned int pend_stride = (unsigned int)(nq * R * 4);   // MFMA flavour: bytes per parked tile
    const bool loader = w >= 4;                     // waves 4-7 stage the tiles
    const bool grpB = w >= 6;                       // ... 6-7 also the per-row aux values
    const int lw = w & 3;
    // fp16: wave 0 multiplies (nq <= 16: the kernel is a streaming kernel); wave 1: selector of queries 0-1, wave 2: of 2-3.
    // float32: waves 0 and 2 compute (even / odd 16-row passes of a tile), wave 1 is the selector of both queries.
    const bool mfma_wave = w == 0 || (VALU && TWOCW && w == 2);
    const int pp0 = VALU && TWOCW ? (w >> 1) : 0;   // first pass of this computing wave
    const bool selector = (w == 1 || (w == 2 && !VALU)) && 2 * (w - 1) < nq;
    const bool requester = w == 3;
    const int64_t G = gridDim.x;
    const int64_t b = blockIdx.x;

    if (tid < 16) ctl[tid] = 0;
#if HDB_FUSED_STAMPS
    bool stamp_wave = w == 0;
#endif
    HDB_STAMP(0);

    // ---- tile sequence: phase A (sample) then phase B (all rows) through ONE ring ------------------
    const char* const Vb = reinterpret_cast<const char*>(a.V);
    const int64_t n_rows = a.n;
    const bool local = PARK && f.local != 0;         // no sample, no exchange: the workgroup's own threshold (see the header)
    constexpr bool SLOTTED = !(METRIC == 2 && sizeof(E) == 2);      // ... and its lists are slotted (FusedArgs::local_slot)
    const int64_t nA = local ? 0 : (f.s_tiles > b ? (f.s_tiles - b + G - 1) / G : 0);
    // Phase B hands out its tiles DYNAMICALLY, in chunks of CH consecutive tiles: the first S chunks of a workgroup are
    // fixed (chunk c -> tiles (c*G + b)*CH ...), every later one comes from a global counter (f.ctl[32], a cache line
    // of its own: ~35 requests per us at N = 10M).  With a static split the slowest workgroup finished 15-25 us after the
    // median one (N = 1.25M .. 10M): that tail is what the counter removes.  Wave 3 requests the chunks and hands the
    // answers over through LDS.
    // The counter is one address: it answers ~70 requests per us, and G workgroups asking for ONE tile per 1.6-us round
    // (160 / us) queue up behind it -- 3 us per tile instead of 1.7 (N = 500k: 109 us for 31 tiles per workgroup).  So
    // requests are always for CHL tiles (half at the very end), matrices of 2..8 requests per workgroup take the first half of
    // their chunks statically, and smaller ones are split statically altogether.
    const int64_t ntiles = a.ntiles;
    // A request covers ~192 KiB of V (4 tiles of 48 KiB, 6 of 32, 12 of 16): tiles of 28-40 KiB stream in 1.0-1.4 us, and four
    // of them per request put G / (4 x 1.1 us) = 58 requests per us on the counter (d=1024: 1.53 us per tile instead of 1.15).
    constexpr int64_t CHL = (4 * 48 * 1024 + STAGE - 1) / STAGE;
    // (local flavour: a fixed split, tile t -> workgroup t mod G, so that nobody exceeds its parking area)
    const int64_t CH = (!local && ntiles >= 2 * CHL * G) ? CHL : 1;      // size of the fixed chunks (and of the large requests)
    const int64_t S = local ? (ntiles + G - 1) / G : ntiles >= 8 * CHL * G ? 1 : ntiles >= 2 * CHL * G ? ntiles / (2 * CHL * G) : (ntiles + G - 1) / G;
    const int64_t dyn0 = S * G * CH;                 // first tile handed out by the counter
    unsigned int* dq = ctl + 4;                      // [2][2] {first tile - dyn0, length} handed over by wave 7
    const unsigned int dq_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(dq);
    int64_t gp = 0;                                  // next sequence position to generate
    int64_t cidx = 0, coff = 0, cbase = 0, clen = CH;   // phase-B generator: chunk number, offset in it, its first tile, its length
    int64_t seen = 0;                                // requester: the counter after its last request
    // -> row0 of sequence position gp (and whether it exists); positions < nA are the sample tiles
    auto gen = [&](int64_t& row0, bool& valid) {
        if (gp < nA) {
            row0 = hdb_tile_index(b + gp * G, f.s_stride) * R; valid = true;
        } else {
            if (coff == 0) {
                if (cidx < S) { cbase = (cidx * G + b) * CH; clen = CH; }
                else {
                    unsigned long long pr;
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pr) : "v"(dq_addr + (unsigned int)(cidx & 1) * 8u) : "memory");
                    cbase = dyn0 + (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)pr);
                    clen = (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(pr >> 32));
                }
            }
            // Request the chunk after this one two rounds before its first tile is generated (the counter answers within a
            // round): a workgroup then never holds more than the chunk it works on, which bounds the finishing skew.
            if (requester && coff == (clen >= 2 ? clen - 2 : 0) && cidx + 1 >= S) {
                // CH tiles per request while plenty are left, half of that for the last rounds of the grid: the slowest workgroups
                // stream ~25 % slower than the fastest, and what they still hold when the counter runs dry is the tail
                const unsigned int want = CH == 1 ? 1u : (ntiles - dyn0 - seen > (CH + CH / 2) * G ? (unsigned int)CH : (unsigned int)(CH / 2));
                unsigned int got = 0u;
                if (lane == 0) got = __hip_atomic_fetch_add(f.ctl + 32, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                got = (unsigned int)__builtin_amdgcn_readfirstlane((int)got);
                seen = (int64_t)got + want;
                const unsigned long long pr = ((unsigned long long)want << 32) | got;
                if (lane == 0) asm volatile("ds_write_b64 %0, %1" :: "v"(dq_addr + (unsigned int)((cidx + 1) & 1) * 8u), "v"(pr) : "memory");
            }
            const int64_t t = cbase + coff;
            valid = t < ntiles; row0 = t * (int64_t)R;
            if (++coff == clen) { coff = 0; ++cidx; }
        }
        ++gp;
    };
    int g_off[NGL];
#pragma unroll
    for (int j = 0; j < NGL; ++j) {
        const int slot = (lw + 4 * j) * 64 + lane;
        const int r = slot / CPR, cpos = slot - r * CPR;
        g_off[j] = r * ROWB + (cpos ^ (r & 15)) * 16;
    }
    auto issue = [&](int64_t row0, int st) {
        if (!loader) return;
        const int64_t last = n_rows - 1 - row0;
        char* sdst = smem + st * STAGE;
        const char* tile_base = Vb + row0 * (int64_t)ROWB;
        if (last >= R - 1) {
#pragma unroll
            for (int j = 0; j < NGL; ++j)
                __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(tile_base + (unsigned int)g_off[j]),
                                                 HDB_LDS_PTR(sdst + (lw + 4 * j) * 1024), 16, 0, 2);
        } else {
#pragma unroll
            for (int j = 0; j < NGL; ++j) {
                const int r = g_off[j] / ROWB;
                const int rr = r <= (int)last ? r : (int)last;
                const unsigned int off = (unsigned int)(g_off[j] + (rr - r) * ROWB);
                __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(tile_base + off), HDB_LDS_PTR(sdst + (lw + 4 * j) * 1024), 16, 0, 2);
            }
        }
        if ((AUX0 || HAS_BIAS) && grpB) {
            const int64_t rr = lane <= last ? lane : last;
            if (AUX0) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(aux0g + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 0) * 64), 4, 0, 0);
            if (HAS_BIAS) __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(a.bias + row0 + rr), HDB_LDS_PTR(auxbuf + (st * 2 + 1) * 64), 4, 0, 0);
        }
    };
    // ---- query preparation (hdb_qprep_kernel, per workgroup): wave q prepares query q.  The query loads are issued
    // BEFORE the first staging so that waiting for them does not wait for the tiles (vmcnt retires in order).
    constexpr int QPL = (D + 63) / 64;
    float qreg[QPL];
    if (w < nq) {
        const float* qv = f.Qraw + (int64_t)w * D;
#pragma unroll
        for (int u = 0; u < QPL; ++u) { const int e = lane + 64 * u; qreg[u] = e < D ? qv[e] : 0.f; }
    }
    int64_t rA, rB, rC = 0; bool vA, vB, vC = false;      // rows / existence of sequence positions p, p+1, p+2
    gen(rA, vA);
    gen(rB, vB);
    if (vA) issue(rA, 0);
    if (vB) issue(rB, 1);
    const unsigned int tsc_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(tsc);
    const unsigned int qpar_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(qpar);
    const unsigned int qlds_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(qlds);
    if (w < nq) {
        if constexpr (METRIC == 3) {                 // centre the query exactly as hdb_qcentre_kernel does (same partial sums, same tree)
            float sm = 0.f;
#pragma unroll
            for (int u = 0; u < QPL; ++u) sm += qreg[u];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
            const float mean = sm / (float)D;
#pragma unroll
            for (int u = 0; u < QPL; ++u) qreg[u] = lane + 64 * u < D ? qreg[u] - mean : 0.f;
        }
        float ss = 0.f, amax = 0.f;
#pragma unroll
        for (int u = 0; u < QPL; ++u) { ss = fmaf(qreg[u], qreg[u], ss); amax = fmaxf(amax, fabsf(qreg[u])); }   // as hdb_qprep_kernel
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); amax = fmaxf(amax, __shfl_xor(amax, o, 64)); }
        const float scale = VALU ? 1.f : hdb_q16_scale(amax);
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
            const int e = lane + 64 * u;
            if constexpr (VALU) {
                if (e < D) hdb_lds_st32(qlds_addr + (unsigned int)(w * D + e) * 4u, qreg[u]);
            } else {
                const _Float16 hv = (_Float16)(qreg[u] * scale);
                if (e < D) hdb_lds_st16(qlds_addr + (unsigned int)(w * D + e) * 2u, (unsigned int)__builtin_bit_cast(unsigned short, hv));
            }
        }
        if (lane == 0) {
            float qinv = (ss == 0.f) ? 1.0f : 1.0f / sqrtf(ss);
            if constexpr (METRIC == 3) { const float sd = sqrtf(ss / (float)D); qinv = (sd == 0.f) ? __builtin_nanf("") : 1.0f / sd; }   // 1/sd_q, NaN for a constant query (:107-111)
            hdb_lds_st32(qpar_addr + (unsigned int)w * 4u, (COSLIKE ? qinv : 1.0f) * (1.f / scale));
            if (METRIC == 2) hdb_lds_st32(qpar_addr + (unsigned int)(3 * HDB_FUSED_MAXQ + w) * 4u, ss);      // ||q||^2
            hdb_lds_st32(qpar_addr + (unsigned int)(HDB_FUSED_MAXQ + w) * 4u, (ss != ss) ? 1.f : 0.f);
            hdb_lds_st32(qpar_addr + (unsigned int)(2 * HDB_FUSED_MAXQ + w) * 4u, __builtin_nanf(""));      // threshold: none yet
        }
    }
    hdb_lds_barrier();
    HDB_STAMP(1);

    // ---- wave 0: B fragments and per-query constants ---------------------------------------------
    const bool q_ok = rl < nq;
    Vec Bq[(VALU || BLDS) ? 1 : KS];
    float qinv_l = 1.f, qsq_l = 0.f;
    // VALU flavour: lane (group g = lane >> 4, l16 = lane & 15) holds chunks l16 + 16 j of every query, and the per-query
    // multipliers / thresholds as wave-uniform values
    const int l16 = lane & 15, g4 = lane >> 4;
    f32x4 qv[VALU ? VQ : 1][VALU ? NJ : 1];
    float qmul[VQ], thr_q[VQ];
#pragma unroll
    for (int q = 0; q < VQ; ++q) { qmul[q] = 1.f; thr_q[q] = INFINITY; }
    if (mfma_wave) {
        if constexpr (VALU) {
#pragma unroll
            for (int q = 0; q < VQ; ++q) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (q < nq) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(qlds_addr + (unsigned int)(q * D) * 4u + (unsigned int)(l16 + 16 * j) * 16u) : "memory");
                    qv[q][j] = v;
                }
                if (q < nq) qmul[q] = hdb_lds_ld32(qpar_addr + (unsigned int)q * 4u);
            }
        } else if constexpr (BLDS) {
            if (q_ok) qinv_l = hdb_lds_ld32(qpar_addr + (unsigned int)rl * 4u);
            if (METRIC == 2 && q_ok) qsq_l = hdb_lds_ld32(qpar_addr + (unsigned int)(3 * HDB_FUSED_MAXQ + rl) * 4u);
        } else {
            const unsigned int src = qlds_addr + (unsigned int)((q_ok ? rl : 0) * D) * 2u + (unsigned int)h * 16u;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Vec v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(src), "i"(CPS * 16 * s));
                Bq[s] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                asm volatile("" : "+v"(Bq[s]));                  // values are final only behind the wait above
                if (!q_ok) Bq[s] = Vec{0, 0, 0, 0, 0, 0, 0, 0};
            }
            if (q_ok) qinv_l = hdb_lds_ld32(qpar_addr + (unsigned int)rl * 4u);
            if (METRIC == 2 && q_ok) qsq_l = hdb_lds_ld32(qpar_addr + (unsigned int)(3 * HDB_FUSED_MAXQ + rl) * 4u);
        }
    }

    const unsigned int smem_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(smem);
    const unsigned int ctl_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(ctl);
    const unsigned int cb_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(cb);
    const unsigned int cbq_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(cbq);
    const unsigned int rd_base = (unsigned int)(rl * CPR * 16);
    const unsigned int hx = (unsigned int)((h ^ (rl & 15)) << 4);

    // LDS candidate list -> global lists.  ONE global atomic per query and flush reserves the slots of all its entries
    // (ranks inside the workgroup come from LDS atomics): every workgroup finishes within a few us of the others, and an
    // atomic per entry on the one counter word of a single query (2048 of them at ~88 per us) held the end of the kernel
    // up by 20 us.
    auto flush = [&]() {
        hdb_lds_barrier();
        const unsigned int ne = ctl[0] < CB ? ctl[0] : CB;
        if (tid < 8) ctl[8 + tid] = 0;               // [8..11] entries per query, [12..15] their first global slot
        hdb_lds_barrier();
        unsigned int rank[(CB + 511) / 512];
#pragma unroll
        for (int u = 0; u < (CB + 511) / 512; ++u) {
            const unsigned int e = tid + 512 * u;
            rank[u] = e < ne ? atomicAdd(&ctl[8 + cbq[e]], 1u) : 0u;
        }
        hdb_lds_barrier();
        if (tid < HDB_FUSED_MAXQ) {
            const unsigned int c = ctl[8 + tid];
            ctl[12 + tid] = c ? atomicAdd(&f.ctl[2 + tid], c) : 0u;
        }
        hdb_lds_barrier();
#pragma unroll
        for (int u = 0; u < (CB + 511) / 512; ++u) {
            const unsigned int e = tid + 512 * u;
            if (e < ne) {
                const unsigned int qe = cbq[e];
                const unsigned int pos = ctl[12 + qe] + rank[u];
                if (pos < f.cap) f.cand[(int64_t)qe * f.cap + pos] = cb[e];
            }
        }
        hdb_lds_barrier();
        if (tid == 0) ctl[0] = 0;
        hdb_lds_barrier();
    };

    // Comparison domain of the filter == domain of the sample scores (identical arithmetic in both phases): the raw
    // dot (x 1/||v|| for cosine) without bias, the finished score with bias.
    float thr_cmp = INFINITY;
    // Entries the LDS list may hold right now.  While parked tiles are being filtered the list grows into the part of the parking
    // area that has been consumed (tile p read -> its bytes are free); only what does not fit goes straight to the global lists,
    // one atomic per entry.  (Round 3 sent every survivor of a parked tile there: on short matrices, where all tiles are parked,
    // 256 workgroups then queued 2-4 k returning atomics on one counter word -- 10-20 us at N = 100k.)
    unsigned int cb_lim = CB;
    auto filter = [&](const Acc (&tv)[RT], int64_t row0) {
        float gm[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) gm[rt] = fmaxf(fmaxf(tv[rt][0], tv[rt][1]), fmaxf(tv[rt][2], tv[rt][3]));
        float m = gm[0];
#pragma unroll
        for (int rt = 1; rt < RT; ++rt) m = fmaxf(m, gm[rt]);
        if (m >= thr_cmp) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if (gm[rt] >= thr_cmp) {
                    const int64_t rowg = row0 + rt * 16 + 4 * h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x = tv[rt][j];
                        if (x >= thr_cmp && q_ok && rowg + j < n_rows && !(HAS_BIAS && x == -INFINITY)) {
                            const float sc = hdb_canon((HAS_BIAS || METRIC == 2) ? x : x * qinv_l);
                            unsigned int pos;
                            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pos) : "v"(ctl_addr), "v"(1u) : "memory");
                            if (pos < cb_lim) {
                                const unsigned long long ent = hdb_pack(sc, (uint32_t)(rowg + j));
                                asm volatile("ds_write_b64 %0, %1\n\tds_write_b16 %2, %3"
                                             :: "v"(cb_addr + pos * 8u), "v"(ent), "v"(cbq_addr + pos * 2u), "v"((unsigned int)rl) : "memory");
                            } else {
                                // (a limited list: give the slot back -- only this wave touches the counter, and the kept slots are the low ones)
                                if (PARK && cb_lim < (unsigned int)CB) asm volatile("ds_add_u32 %0, %1" :: "v"(ctl_addr), "v"(0xFFFFFFFFu) : "memory");
                                const unsigned int gpos = atomicAdd(&f.ctl[2 + rl], 1u);
                                if (gpos < f.cap) f.cand[(int64_t)rl * f.cap + gpos] = hdb_pack(sc, (uint32_t)(rowg + j));
                            }
                        }
                    }
                }
            }
        }
    };

    // VALU flavour: the comparable values of a tile, one row per owning lane ((l16 & 3) == 0) and pass
    float pend[VALU ? NPW : 1][VQ];
    auto filter_valu = [&](const float (&tv)[VALU ? NPW : 1][VQ], int64_t row0) {
        const int u_own = hdb_owned_row(l16);
#pragma unroll
        for (int k2 = 0; k2 < (VALU ? NPW : 1); ++k2) {
            const int pp = (TWOCW ? 2 : 1) * k2 + pp0;
            const int64_t row = row0 + 16 * pp + 4 * g4 + u_own;
#pragma unroll
            for (int q = 0; q < VQ; ++q) {
                const float x = tv[k2][q];
                if (q < nq && (l16 & 3) == 0 && x >= thr_q[q] && row < n_rows && !(HAS_BIAS && x == -INFINITY)) {
                    const float sc = hdb_canon((HAS_BIAS || METRIC == 2) ? x : x * qmul[q]);
                    unsigned int pos;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pos) : "v"(ctl_addr), "v"(1u) : "memory");
                    const unsigned long long ent = hdb_pack(sc, (uint32_t)row);
                    if (pos < CB) {
                        asm volatile("ds_write_b64 %0, %1\n\tds_write_b16 %2, %3"
                                     :: "v"(cb_addr + pos * 8u), "v"(ent), "v"(cbq_addr + pos * 2u), "v"((unsigned int)q) : "memory");
                    } else {
                        const unsigned int gpos = atomicAdd(&f.ctl[2 + q], 1u);
                        if (gpos < f.cap) f.cand[(int64_t)q * f.cap + gpos] = ent;
                    }
                }
            }
        }
    };

    // ---- selector state: the M largest sample scores of this wave's query, as orderable keys, lane r < M holds one
    uint32_t keep[2] = {0u, 0u};                     // key 0 sorts below every float, -inf included
    bool thr_final[2] = {false, false};              // every workgroup's sample went into this query's threshold
    bool have_thr[2] = {false, false};               // some usable threshold of this query is in qpar
    int64_t ref_at[2] = {-1, -1};                    // round (relative to nA) of the next refinement sweep
    int ref_n[2] = {0, 0};
    bool poll_issued = false;                        // workgroups > 0: an LDS-DMA poll of the threshold words is in flight
    uint32_t poll_key[2] = {0u, 0u};
    bool gave_up = false;
    unsigned long long sweep_t0 = 0ull;
    auto merge_tile = [&](uint32_t& keep, unsigned int src64_addr) {      // 64 new scores, one per lane
        uint32_t cur = hdb_f2key(hdb_canon(hdb_lds_ld32(src64_addr + (unsigned int)lane * 4u)));
        uint32_t out = 0u;
#pragma unroll
        for (int r = 0; r < M; ++r) {
            uint32_t v = max(cur, keep);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
            const unsigned long long who = __ballot(max(cur, keep) == v);
            const int first = (int)__ffsll((long long)who) - 1;
            if (lane == first) { if (cur >= keep) cur = 0u; else keep = 0u; }
            if (lane == r) out = v;
        }
        keep = out;
    };
    auto spin_expired = [&](unsigned long long t0) {
        return (unsigned long long)__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)f.timeout_ticks;
    };

    const int chk_shift = a.ntiles >= 65536 ? 4 : 0;
    const int64_t chk_mask = (1 << chk_shift) - 1;
    Acc acc[RT];
    int64_t row0_prev = 0;
    bool have_prev = false;                          // wave 0: acc holds an unfiltered phase-B tile
    int thr_reads = 24;                              // wave 0 re-reads the threshold for the rounds in which refinements can still raise it
    bool thr_known = false;                          // every query of this call has a threshold in qpar
    int npend = 0;                                   // parked filter tiles
    const unsigned int pbuf_addr = (unsigned int)(uintptr_t)HDB_LDS_PTR(pbuf);
    auto park_row0 = [&](int slot, int64_t row0) {   // first row of parked tile `slot` (tsc is free after the sample phase)
        if (lane == 0) asm volatile("ds_write_b64 %0, %1" :: "v"(tsc_addr + (unsigned int)slot * 8u), "v"((unsigned long long)row0) : "memory");
    };
    auto parked_row0 = [&](int slot) {
        unsigned long long v;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(tsc_addr + (unsigned int)slot * 8u) : "memory");
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(v >> 32));
        return (int64_t)(((unsigned long long)hi << 32) | lo);
    };
    int st_cur = 0;
    for (int64_t i = 0;; ++i) {                      // the round after the last tile drains the deferred work
        const bool tile = vA;
        if (tile && loader) {
            if (!vB) hdb_wait_vmcnt<0>();
            else if (grpB) hdb_wait_vmcnt<NLOADB>();
            else hdb_wait_vmcnt<NLOADA>();
        }
        const bool chk = tile && i >= nA && ((i - nA) & chk_mask) == chk_mask;
        const int chk_slot = 1 + (int)(((i - nA) >> chk_shift) & 1);
        if (chk && tid == 0) ctl[chk_slot] = (ctl[0] >= CB / 4) ? 1u : 0u;
        hdb_lds_barrier();                           // tile i is in LDS; everyone is done with tile i-1; tsc/qpar/dq hand-offs
        const int st_next2 = st_cur == 0 ? 2 : st_cur - 1;
        if (vB) gen(rC, vC); else vC = false;        // positions past the end are never generated (no stray requests)
        if (vC) issue(rC, st_next2);
        if (chk && ctl[chk_slot]) flush();

        // ---- selectors: sample bookkeeping one tile behind wave 0, then the exchange ----------------------
        // Any M-th largest over a SUBSET of the published sample values is a lower bound of the final threshold, so a
        // sweep that finds only part of the granules tagged with this call's epoch already yields a safe (merely less
        // selective) threshold: the filter pass starts on it and later sweeps raise it.  Nothing waits for the slowest
        // workgroup, and a grid that is not fully resident cannot block (see the header).
        if (selector && !local) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * (w - 1) + qq;
                if (q >= nq) continue;
                if (i >= 1 && i <= nA) merge_tile(keep[qq], tsc_addr + (unsigned int)(((int)((i - 1) & 1) * HDB_FUSED_MAXQ + q) * 64) * 4u);
                if (i == nA) {                       // publish M granules {epoch, key}: the data is the flag
                    hdb_gu64* mine = (hdb_gu64*)(f.gran + (b * HDB_FUSED_GRAN_PER_WG + q * M));
                    if (lane < M) __hip_atomic_store(mine + lane, ((unsigned long long)f.epoch << 32) | keep[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if HDB_FUSED_STAMPS
                    stamp_wave = w == 1 && qq == 0;
#endif
                    HDB_STAMP(2);
                    sweep_t0 = __builtin_amdgcn_s_memrealtime();
                }
            }
            // First sweep right after publishing, then one per round until one is usable: wave 0 parks the scores of up to
            // HDB_FUSED_PEND filter tiles meanwhile, so the stream does not stop for the ~10 us the exchange takes; only when
            // that budget is spent (or the stream ends) does the selector hold the round until a threshold exists.  Two
            // refinement sweeps follow 3 and 7 rounds after the first usable one.
            const int64_t di = i - nA;
            if (di >= 0) {
                // workgroups > 0: the two threshold words of this wave's queries as the LDS-DMA poll of the previous round left
                // them in LDS (16 bytes, one lane, one request: a word cannot tear)
                unsigned long long pollw[2] = {0ull, 0ull};          // tag 0 is never an epoch
                const unsigned int poll_addr = tsc_addr + 1024u + (unsigned int)(w - 1) * 16u;
                if (b != 0 && poll_issued) {
                    asm volatile("s_waitcnt vmcnt(0)\n\tds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(pollw[0]), "=&v"(pollw[1]) : "v"(poll_addr) : "memory");
                    poll_issued = false;
                }
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int q = 2 * (w - 1) + qq;
                    if (q >= nq || thr_final[qq] || gave_up) continue;
                    hdb_gu64* const word = (hdb_gu64*)(f.ctl + 64 + 2 * q);       // {final << 31 | epoch, key}: workgroup 0's result (a cache line of their own)
                    if (b != 0) {
                        // Every workgroup but the first POLLS one word per query instead of sweeping the granules itself: a
                        // sweep is ~3 us of loads behind the CU's own staging traffic and stretched each round it ran in from
                        // 1.7 to 3.2 us (five of them per call).  The poll is an LDS-DMA load issued in one round and looked at
                        // in the next (pollw, see below), so a round never waits for it; only when the parking budget is
                        // spent does the wave spin on the word itself.
                        const bool first = !have_thr[qq];
                        const bool may_skip = PARK && first && di < pend_max && vC;
                        unsigned long long v = pollw[qq];
                        for (;;) {
                            if (((uint32_t)(v >> 32) & 0x7FFFFFFFu) == f.epoch) {
                                const uint32_t key = (uint32_t)v;
                                if (first || key != poll_key[qq]) {
                                    if (lane == 0) hdb_lds_st32(qpar_addr + (unsigned int)(2 * HDB_FUSED_MAXQ + q) * 4u, key == 0u ? -INFINITY : hdb_key2f(key));
                                    poll_key[qq] = key;
                                }
                                if (first) HDB_STAMP(3);
                                have_thr[qq] = true;
                                thr_final[qq] = (v >> 63) != 0ull;
                                break;
                            }
                            if (may_skip || !first) break;
                            if (spin_expired(sweep_t0)) {
                                // giving up ends the exchange for EVERY query of the call (the loop skips this wave's other
                                // query from here on): a threshold left at NaN would keep wave 0 parking for good
                                gave_up = true;
                                if (lane < nq) hdb_lds_st32(qpar_addr + (unsigned int)(2 * HDB_FUSED_MAXQ + lane) * 4u, INFINITY);
                                if (lane == 0) atomicOr(&f.ctl[1], 1u);
                                break;
                            }
                            __builtin_amdgcn_s_sleep(2);
                            v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        continue;
                    }
                    const bool first = PARK ? !have_thr[qq] : di == 0;
                    if (PARK ? (!first && di != ref_at[qq]) : (di != 0 && di != 3 && di != 7)) continue;
                    const bool may_skip = PARK && first && di < pend_max && vC;
                    constexpr int WPL = 64 / M;      // lane l sweeps granule (l % M) of workgroups l / M, l / M + WPL, ...
                    const int gi = lane % M, wg0 = lane / M;
                    // usable = at least 16 workgroups WITH sample rows have answered (their 8 x 16 best values put the
                    // M-th largest at ~1 % of the rows: a few extra survivors for a round or two), or everybody has
                    const int samp_wgs = (int)(f.s_tiles < G ? f.s_tiles : G);   // workgroups 0 .. samp_wgs-1 have sample tiles: only they are swept
                    const int need_wgs = samp_wgs < 16 ? samp_wgs : 16;
                    for (;;) {
                        bool ok = true;
                        int nvalid = 0;
                        uint32_t lmax[M];            // per lane: the M largest of ITS valid granules (sorted descending)
#pragma unroll
                        for (int r = 0; r < M; ++r) lmax[r] = 0u;
                        constexpr int SB = VALU ? 16 : 8;       // loads in flight per lane (fp16 flavour: wave 0 holds the B fragments, registers are short)
                        for (int base = wg0; base < samp_wgs; base += SB * WPL) {
                            unsigned long long x[SB];
#pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                const int wg = base + u * WPL;
                                x[u] = wg < samp_wgs ? __hip_atomic_load((hdb_gu64*)(f.gran + (wg * HDB_FUSED_GRAN_PER_WG + q * M + gi)),
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                              : 0ull;    // tag 0 is never an epoch
                            }
#pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                const bool tagged = (uint32_t)(x[u] >> 32) == f.epoch;
                                uint32_t v = tagged ? (uint32_t)x[u] : 0u;
                                if (base + u * WPL < samp_wgs) { ok &= tagged; nvalid += v != 0u ? 1 : 0; }
#pragma unroll
                                for (int r = 0; r < M; ++r) { const uint32_t hi = max(lmax[r], v); v = min(lmax[r], v); lmax[r] = hi; }
                            }
                        }
                        const bool all = __all(ok);
                        // sample workgroups seen = tagged, non-empty granules of position 0 (the largest value of a workgroup)
                        int cnt0 = gi == 0 ? nvalid : 0;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) cnt0 += __shfl_xor(cnt0, o, 64);
                        if (all || cnt0 >= need_wgs) {
                            uint32_t kth = 0u;       // M rounds: the wave-wide maximum of the lane heads; its owner pops its head
#pragma unroll
                            for (int r = 0; r < M; ++r) {
                                uint32_t v = lmax[0];
#pragma unroll
                                for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
                                const unsigned long long who = __ballot(lmax[0] == v);
                                if (lane == (int)__ffsll((long long)who) - 1) {
#pragma unroll
                                    for (int t = 0; t + 1 < M; ++t) lmax[t] = lmax[t + 1];
                                    lmax[M - 1] = 0u;
                                }
                                kth = v;
                            }
                            thr_final[qq] = all;
                            if (first) HDB_STAMP(3);
                            if constexpr (PARK) {
                                have_thr[qq] = true;
                                ref_at[qq] = ref_n[qq] == 0 ? di + 3 : ref_n[qq] == 1 ? di + 4 : -1;
                                ++ref_n[qq];
                            }
                            const unsigned long long pubw = ((unsigned long long)(f.epoch | (all ? 0x80000000u : 0u)) << 32) | kth;
                            if (lane == 0) {
                                const float thr = kth == 0u ? -INFINITY : hdb_key2f(kth);
                                hdb_lds_st32(qpar_addr + (unsigned int)(2 * HDB_FUSED_MAXQ + q) * 4u, thr);
                                f.thr_out[q] = thr;              // the highest threshold anybody filters with (read by the last workgroup)
                                // for everybody else; bit 31 of the tag carries "final" (epochs are 31 bits wide)
                                __hip_atomic_store(word, pubw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            if (lane < HDB_FUSED_POLLS)      // ... and the copies the pollers read, one per round
                                __hip_atomic_store((hdb_gu64*)(reinterpret_cast<char*>(f.ctl) + 512 + lane * 128 + (w - 1) * 16 + qq * 8),
                                                   pubw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        if (may_skip) break;             // try again next round; this round's tile gets parked
                        if (spin_expired(sweep_t0)) {    // too few workgroups answered within the timeout: give up (host falls back)
                            gave_up = true;                  // for every query of the call, see above
                            if (lane < nq) hdb_lds_st32(qpar_addr + (unsigned int)(2 * HDB_FUSED_MAXQ + lane) * 4u, INFINITY);
                            if (lane == 0) atomicOr(&f.ctl[1], 1u);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                // next poll (copy number `di`): while a final threshold is still to come, for at most HDB_FUSED_POLLS rounds
                if (b != 0 && !gave_up && di < HDB_FUSED_POLLS && ((2 * (w - 1) < nq && !thr_final[0]) || (2 * (w - 1) + 1 < nq && !thr_final[1]))) {
                    if (lane == 0)
                        __builtin_amdgcn_global_load_lds(HDB_GLOBAL_PTR(reinterpret_cast<const char*>(f.ctl) + 512 + di * 128 + (w - 1) * 16),
                                                         HDB_LDS_PTR(tsc + 256 + 4 * (w - 1)), 16, 0, 16 /* sc1: agent scope */);
                    poll_issued = true;
                }
#if HDB_FUSED_STAMPS
                if ((!PARK && b == 0) || have_thr[0] || gave_up) stamp_wave = false;
#endif
            }
        }

        // ---- wave 0: multiply tile i; epilogue of a sample tile at once, of a filter tile one round later ----
        if constexpr (VALU) {
        if (mfma_wave) {
            if (have_prev) {                         // deferred epilogue of tile i-1 (phase B)
                if (!thr_known || thr_reads > 0) {   // written by the selectors before this round's barrier; later sweeps raise it
                    float tq[VQ];
                    bool kn = true;
#pragma unroll
                    for (int q = 0; q < VQ; ++q) {
                        tq[q] = q < nq ? hdb_lds_ld32(qpar_addr + (unsigned int)(2 * HDB_FUSED_MAXQ + q) * 4u) : INFINITY;
                        kn = kn && tq[q] == tq[q];   // NaN: no threshold yet
                    }
                    if (kn) {
#pragma unroll
                        for (int q = 0; q < VQ; ++q) if (q < nq) thr_q[q] = tq[q];
                        thr_known = true;
                        --thr_reads;
                    }
                }
                const int u_own_p = hdb_owned_row(l16);
                if (!thr_known && npend >= pend_max) {
                    // the parking area is full and no threshold has come (the selector holds the round until one exists, so
                    // this means the exchange was abandoned): drop the tile under the abort word -- the host re-runs the call
                    if (lane == 0) atomicOr(&f.ctl[1], 1u);
                } else if (!thr_known) {             // parking: keep the comparable values of this tile, filter them later
#pragma unroll
                    for (int q = 0; q < VQ; ++q)
#pragma unroll
                        for (int k2 = 0; k2 < NPW; ++k2)
                            if (q < nq && (l16 & 3) == 0)
                                hdb_lds_st32(pbuf_addr + (unsigned int)((npend * VQ + q) * R + 16 * ((TWOCW ? 2 : 1) * k2 + pp0) + 4 * g4 + u_own_p) * 4u, pend[k2][q]);
                    if (w == 0) park_row0(npend, row0_prev);
                    ++npend;
                } else {
#pragma unroll 1
                    for (int p = 0; p < npend; ++p) {
                        float tv[NPW][VQ];
#pragma unroll
                        for (int q = 0; q < VQ; ++q)
#pragma unroll
                            for (int k2 = 0; k2 < NPW; ++k2)
                                tv[k2][q] = q < nq ? hdb_lds_ld32(pbuf_addr + (unsigned int)((p * VQ + q) * R + 16 * ((TWOCW ? 2 : 1) * k2 + pp0) + 4 * g4 + u_own_p) * 4u) : -INFINITY;
                        filter_valu(tv, parked_row0(p));
                    }
                    npend = 0;
                    filter_valu(pend, row0_prev);
                }
                have_prev = false;
            }
            if (tile) {
                const int64_t row0 = rA;
                const char* tbase = smem + st_cur * STAGE;
                const float* ax0 = auxbuf + (st_cur * 2 + 0) * 64;
                const float* ax1 = auxbuf + (st_cur * 2 + 1) * 64;
                const int u_own = hdb_owned_row(l16);
#pragma unroll
                for (int k2 = 0; k2 < NPW; ++k2) {
                    const int pp = (TWOCW ? 2 : 1) * k2 + pp0;
                    // rows 16 pp + 4 g4 + u (u = 0..3); chunk c of row r sits at ((c ^ (r & 15)) << 4) of its row image
                    float accv[4][VQ];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int q = 0; q < VQ; ++q) accv[u][q] = 0.f;
                    constexpr int JB = NJ % 3 == 0 ? 3 : (NJ % 2 == 0 ? 2 : 1);       // chunks in flight per row and step (registers)
#pragma unroll
                    for (int j0 = 0; j0 < NJ; j0 += JB) {
                        __builtin_amdgcn_sched_barrier(0);     // keep hipcc from hoisting the next step's 12 reads over this one (spills)
                        f32x4 raw[JB][4];
#pragma unroll
                        for (int j = 0; j < JB; ++j)
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int r = 16 * pp + 4 * g4 + u;
                                raw[j][u] = *reinterpret_cast<const f32x4*>(tbase + r * ROWB + (((l16 + 16 * (j0 + j)) ^ (r & 15)) << 4));
                            }
#pragma unroll
                        for (int j = 0; j < JB; ++j)
#pragma unroll
                            for (int q = 0; q < VQ; ++q)
#pragma unroll
                                for (int u = 0; u < 4; ++u)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        if constexpr (METRIC == 2) { const float df = raw[j][u][e] - qv[q][j0 + j][e]; accv[u][q] = fmaf(df, df, accv[u][q]); }
                                        else accv[u][q] = fmaf(raw[j][u][e], qv[q][j0 + j][e], accv[u][q]);
                                    }
                    }
                    const int rloc = 16 * pp + 4 * g4 + u_own;             // the row this lane owns after the butterfly
                    const float a0 = AUX0 ? ax0[rloc] : 1.f;
                    const float b0 = HAS_BIAS ? ax1[rloc] : 0.f;
#pragma unroll
                    for (int q = 0; q < VQ; ++q) {
                        const float dot = hdb_rows4_sum(accv[0][q], accv[1][q], accv[2][q], accv[3][q], l16);
                        if constexpr (METRIC == 2) {          // hdb_emit's expression (hdb_scan.hip): 1 / (1 + ||v - q||), then the bias
                            const float sim = 1.f / (1.f + sqrtf(dot));
                            pend[k2][q] = HAS_BIAS ? sim + b0 : sim;
                            continue;
                        }
                        const float rawv = COSLIKE ? dot * a0 : dot;
                        // rounding steps of hdb_emit (hdb_scan.hip): product rounded, THEN the bias added -- the empty asm keeps
                        // hipcc from contracting the two into one fma, so both pipelines return bit-identical scores
                        float scaled = rawv * qmul[q];
                        asm volatile("" : "+v"(scaled));
                        pend[k2][q] = HAS_BIAS ? scaled + b0 : rawv;
                    }
                }
                if (i < nA) {                        // sample tile: scores to the selectors (read after the next barrier)
                    if ((l16 & 3) == 0) {
#pragma unroll
                        for (int q = 0; q < VQ; ++q) {
                            if (q < nq) {
                                const unsigned int dst = tsc_addr + (unsigned int)(((int)(i & 1) * HDB_FUSED_MAXQ + q) * 64) * 4u;
#pragma unroll
                                for (int k2 = 0; k2 < NPW; ++k2) hdb_lds_st32(dst + (unsigned int)(16 * ((TWOCW ? 2 : 1) * k2 + pp0) + 4 * g4 + u_own) * 4u, pend[k2][q]);
                            }
                        }
                    }
                    if (R < 64 && w == 0 && lane < 64 - R) {   // shorter tiles: the rest of the 64-entry row is -inf
#pragma unroll
                        for (int q = 0; q < VQ; ++q)
                            if (q < nq) hdb_lds_st32(tsc_addr + (unsigned int)((((int)(i & 1) * HDB_FUSED_MAXQ + q) * 64) + R + lane) * 4u, -INFINITY);
                    }
                } else {
                    row0_prev = row0;
                    have_prev = true;
#if HDB_FUSED_STAMPS
                    { const int64_t dj = i - nA + 1; if (dj == 1) HDB_STAMP(11); else if (dj == 4) HDB_STAMP(12); else if (dj == 8) HDB_STAMP(13); else if (dj == 16) HDB_STAMP(14); else if (dj == 32) HDB_STAMP(15); }
#endif
                }
            }
        }
        } else {
        if (mfma_wave) {
            if (have_prev) {                         // deferred epilogue of tile i-1 (phase B)
                if ((PARK && !thr_known) || thr_reads > 0) {   // written by the selectors before this round's barrier; later sweeps raise it
                    const float t = hdb_lds_ld32(qpar_addr + (unsigned int)(2 * HDB_FUSED_MAXQ + (q_ok ? rl : 0)) * 4u);
                    if (!PARK || __all(t == t)) {    // NaN: some query has no threshold yet
                        thr_cmp = q_ok ? t : INFINITY;
                        thr_known = true;
                        --thr_reads;
                    }
                }
                const unsigned int pslot = (unsigned int)(rl * R + 4 * h) * 4u;
                if (PARK && !thr_known && npend >= pend_max) {
                    if (lane == 0) atomicOr(&f.ctl[1], 1u);      // parking area full, exchange abandoned: drop the tile under the abort word
                } else if (PARK && !thr_known) {     // parking: keep the comparable values of this tile, filter them later
                    if (q_ok) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) hdb_lds_st128(pbuf_addr + (unsigned int)npend * pend_stride + pslot + (unsigned int)rt * 64u, acc[rt]);
                    }
                    park_row0(npend, row0_prev);
                    ++npend;
                } else {
                    if (PARK && npend > 0) {
#pragma unroll 1
                        for (int p = 0; p < npend; ++p) {
                            Acc tv[RT];
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt) {
                                f32x4 v = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                                if (q_ok) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(pbuf_addr + (unsigned int)p * pend_stride + pslot + (unsigned int)rt * 64u) : "memory");
                                tv[rt] = v;
                            }
                            // the parked scores sit where the candidate list grows: it may use the bytes of tiles 0 .. p, now in registers
                            const unsigned int freed = (unsigned int)(p + 1) * pend_stride / 8u;
                            cb_lim = freed < (unsigned int)CB ? freed : (unsigned int)CB;
                            filter(tv, parked_row0(p));
                        }
                        cb_lim = CB;
                        npend = 0;
                    }
                    filter(acc, row0_prev);
                }
                have_prev = false;
            }
            if (tile) {
                const int64_t row0 = rA;
                const unsigned int sb_addr = smem_addr + (unsigned int)(st_cur * STAGE) + rd_base;
                constexpr int PF = 3;
                Vec abuf[PF + 1][RT];
                auto fetch = [&](int s, Vec (&dst)[RT]) {
                    const unsigned int ad = sb_addr + ((unsigned int)(16 * CPS * s) ^ hx);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(ad));
                    if constexpr (RT > 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(ad), "i"(MF * CPR * 16));
                    if constexpr (RT > 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[2]) : "v"(ad), "i"(2 * MF * CPR * 16));
                    if constexpr (RT > 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[3]) : "v"(ad), "i"(3 * MF * CPR * 16));
                };
                auto wait_frag = [&](int pend, Vec (&fr)[RT]) {
                    static_assert(RT == 1 || RT == 2 || RT == 4, "RT");
#define HDB_WAITF(N)                                                                                                  \
                    do {                                                                                               \
                        if constexpr (RT == 1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(fr[0]));                 \
                        else if constexpr (RT == 2) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(fr[0]), "+v"(fr[1])); \
                        else asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3])); \
                    } while (0)
                    const int cnt = pend * RT;
                    if (cnt >= 12) HDB_WAITF(12); else if (cnt == 8) HDB_WAITF(8); else if (cnt == 6) HDB_WAITF(6);
                    else if (cnt == 4) HDB_WAITF(4); else if (cnt == 3) HDB_WAITF(3); else if (cnt == 2) HDB_WAITF(2);
                    else if (cnt == 1) HDB_WAITF(1); else HDB_WAITF(0);
#undef HDB_WAITF
                };
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[rt][e] = 0.f;
                if constexpr (BLDS) {
                    // query fragments from LDS, one read per k-step next to the row fragment (RT == 1): two LDS reads per step,
                    // PFB steps in flight (lgkmcnt counts to 15: 7 steps at most) -- with 3 steps the LDS latency under staging
                    // traffic was exposed at every step); lanes of columns without a query read query 0's fragment
                    constexpr int PFB = 7;
                    const unsigned int bsrc = qlds_addr + (unsigned int)((q_ok ? rl : 0) * D) * 2u + (unsigned int)h * 16u;
                    Vec abl[PFB + 1], bbl[PFB + 1];
                    auto fetch_ab = [&](int s, Vec& da, Vec& db) {
                        const unsigned int ad = sb_addr + ((unsigned int)(16 * CPS * s) ^ hx);
                        asm volatile("ds_read_b128 %0, %1" : "=v"(da) : "v"(ad));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(db) : "v"(bsrc), "i"(CPS * 16 * (s < KS ? s : 0)));
                    };
#pragma unroll
                    for (int s = 0; s < PFB && s < KS; ++s) fetch_ab(s, abl[s % (PFB + 1)], bbl[s % (PFB + 1)]);
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        if (s + PFB < KS) fetch_ab(s + PFB, abl[(s + PFB) % (PFB + 1)], bbl[(s + PFB) % (PFB + 1)]);
                        const int pend = (KS - 1 - s) < PFB ? (KS - 1 - s) : PFB;      // steps still in flight behind step s: 2 reads each
                        Vec& af = abl[s % (PFB + 1)];
                        Vec& bf = bbl[s % (PFB + 1)];
                        if (pend == 7) asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(af), "+v"(bf));
                        else if (pend == 6) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(af), "+v"(bf));
                        else if (pend == 5) asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(af), "+v"(bf));
                        else if (pend == 4) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(af), "+v"(bf));
                        else if (pend == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af), "+v"(bf));
                        else if (pend == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af), "+v"(bf));
                        else if (pend == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af), "+v"(bf));
                        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af), "+v"(bf));
                        acc[0] = Shape::mma(af, bf, acc[0]);     // columns without a query repeat query 0 and are never looked at
                    }
                } else {
#pragma unroll
                for (int s = 0; s < PF && s < KS; ++s) fetch(s, abuf[s % (PF + 1)]);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (s + PF < KS) fetch(s + PF, abuf[(s + PF) % (PF + 1)]);
                    const int pend = (KS - 1 - s) < PF ? (KS - 1 - s) : PF;
                    wait_frag(pend, abuf[s % (PF + 1)]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt] = Shape::mma(abuf[s % (PF + 1)][rt], Bq[s], acc[rt]);
                }
                }
                // comparable values, in place
                if (METRIC != 0 || HAS_BIAS) {
                    const float* ax0 = auxbuf + (st_cur * 2 + 0) * 64;
                    const float* ax1 = auxbuf + (st_cur * 2 + 1) * 64;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int rl0 = rt * 16 + 4 * h;
                        float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (AUX0) av = *reinterpret_cast<const float4*>(ax0 + rl0);
                        if (HAS_BIAS) bv = *reinterpret_cast<const float4*>(ax1 + rl0);
                        const float aj[4] = {av.x, av.y, av.z, av.w};
                        const float bj[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float dot = acc[rt][j];
                            if constexpr (METRIC == 2) {     // the batched kernel's expression (hdb_mfma_kernel.h): bit-identical scores
                                const float d2 = fmaxf(fmaf(-2.f * qinv_l, dot, aj[j] + qsq_l), 0.f);
                                acc[rt][j] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_sqrtf(d2)) + (HAS_BIAS ? bj[j] : 0.f);
                            } else {
                                const float raw = COSLIKE ? dot * aj[j] : dot;
                                acc[rt][j] = HAS_BIAS ? fmaf(raw, qinv_l, bj[j]) : raw;
                            }
                        }
                    }
                }
                if (i < nA) {                        // sample tile: scores to the selectors (read after the next barrier)
                    if (q_ok) {
                        const unsigned int dst = tsc_addr + (unsigned int)(((int)(i & 1) * HDB_FUSED_MAXQ + rl) * 64) * 4u;
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) hdb_lds_st128(dst + (unsigned int)(rt * 16 + 4 * h) * 4u, acc[rt]);
                        if (R < 64)                  // shorter tiles: the rest of the 64-entry row is -inf
                            for (int e = R + 4 * h; e < 64; e += 16) hdb_lds_st128(dst + (unsigned int)e * 4u, f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY});
                    }
                } else {
                    row0_prev = row0;
                    have_prev = true;
#if HDB_FUSED_STAMPS
                    { const int64_t dj = i - nA + 1; if (dj == 1) HDB_STAMP(11); else if (dj == 4) HDB_STAMP(12); else if (dj == 8) HDB_STAMP(13); else if (dj == 16) HDB_STAMP(14); else if (dj == 32) HDB_STAMP(15); }
#endif
                }
            }
        }
        }
        st_cur = st_cur == 2 ? 0 : st_cur + 1;
        rA = rB; vA = vB; rB = rC; vB = vC;
        if (!tile) break;
    }
    HDB_STAMP(4);
#if HDB_FUSED_STAMPS
    if (tid == 0) hdb_fused_stamps[16 * blockIdx.x + 7] = (unsigned long long)(gp - nA);      // positions generated in phase B
#endif
    if (!local) flush();
    else {
        // ---- local flavour: every tile of this workgroup is parked.  Wave q: threshold b_w of query q, then its rows ----
        if (w == 0 && lane == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(ctl_addr + 32u), "v"((unsigned int)npend) : "memory");
        hdb_lds_barrier();
        HDB_STAMP(2);
        if (w < nq) {
            const int q = w;
            unsigned int np_u;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(np_u) : "v"(ctl_addr + 32u) : "memory");
            const int nvals = __builtin_amdgcn_readfirstlane((int)np_u) * R;
            const int tstride = (VALU ? VQ : nq) * R;                 // floats per parked tile
            const float qm = hdb_lds_ld32(qpar_addr + (unsigned int)q * 4u);
            // this lane's share of the workgroup's scores of query q, in registers: flat index fi = lane + 64 u -> tile fi / R, row fi % R
            // (plain LDS reads: the stream is over, nothing is in flight that they could drain)
            constexpr int NI = 16;                                     // 16 tiles x 64 rows / 64 lanes
            const float* pq = pbuf + q * R;
            const unsigned long long* prow0 = reinterpret_cast<const unsigned long long*>(tsc);
            float xs[NI]; uint32_t ks[NI], rws[NI];
            uint32_t cur = 0u;
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int fi = lane + 64 * u;
                ks[u] = 0u; xs[u] = 0.f; rws[u] = 0u;
                if (fi < nvals) {
                    const int p = fi / R, r = fi - p * R;
                    const float x = pq[p * tstride + r];
                    const int64_t row = (int64_t)prow0[p] + r;
                    // key 0: not a candidate (past the last row, or excluded by the row mask)
                    if (row < n_rows && !(HAS_BIAS && x == -INFINITY)) ks[u] = hdb_f2key(hdb_canon(x));
                    xs[u] = x; rws[u] = (uint32_t)row;
                }
                cur = max(cur, ks[u]);
            }
            // the local_m-th largest LANE maximum: at least local_m rows reach it (fewer lanes with rows: 0 = everything goes)
            uint32_t thr_key = 0u;
            for (uint32_t r = 0; r < f.local_m; ++r) {
                const uint32_t v = hdb_wave_max_dpp(cur);
                thr_key = v;
                if (v == 0u) break;
                const unsigned long long who = __ballot(cur == v);
                if (lane == (int)__ffsll((long long)who) - 1) cur = 0u;
            }
            uint32_t cnt = 0u;
#pragma unroll
            for (int u = 0; u < NI; ++u) cnt += (uint32_t)__popcll(__ballot(ks[u] != 0u && ks[u] >= thr_key));
            // slotted lists: this workgroup's own region of the list, no reservation (the last workgroup reads the count below);
            // fp16 euclidean keeps a compact list (one returning atomic) for the near-duplicate re-score of hdb_finalize_body
            uint32_t base = (uint32_t)b * f.local_slot;
            const uint32_t lim = SLOTTED ? base + f.local_slot : f.cap;
            if constexpr (!SLOTTED) {
                base = 0u;
                if (lane == 0 && cnt) base = atomicAdd(&f.ctl[2 + q], cnt);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            }
            uint32_t run = 0u;
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const bool pass = ks[u] != 0u && ks[u] >= thr_key;
                const unsigned long long bal = __ballot(pass);
                if (pass) {
                    const uint32_t pos = base + run + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    const float sc = hdb_canon((HAS_BIAS || METRIC == 2) ? xs[u] : xs[u] * qm);
                    if (pos < lim) f.cand[(int64_t)q * f.cap + pos] = hdb_pack(sc, rws[u]);
                }
                run += (uint32_t)__popcll(bal);
            }
            // b_w as a key of the SCORE domain (what the entries carry): rows this workgroup did not emit score at or below it
            if (lane == 0) {
                const float xt = hdb_key2f(thr_key);
                const uint32_t bkey = thr_key == 0u ? 0u : hdb_f2key(hdb_canon((HAS_BIAS || METRIC == 2) ? xt : xt * qm));
                __hip_atomic_store((hdb_gu64*)(f.gran + (b * HDB_FUSED_GRAN_PER_WG + q)), ((unsigned long long)cnt << 32) | bkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        HDB_STAMP(3);
    }

    // ---- finish: drain, release, ticket; the last workgroup finalizes every query -------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int ticket = __hip_atomic_fetch_add(f.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = ticket == (unsigned int)G - 1u;
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ctl[3] = last ? 1u : 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    HDB_STAMP(5);
    if (ctl[3]) {
        unsigned long long* fbuf = reinterpret_cast<unsigned long long*>(smem);     // the ring is free now
        const unsigned int aborted = __hip_atomic_load(f.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int qnan_bits = 0u;
        // Workgroups filter with thresholds that only rise (provisional, then refined): the list may hold rows that only
        // some of them collected.  What is guaranteed complete is everything at or above the LAST threshold workgroup 0
        // published (nobody used a higher one), so the top-k is exact iff at least kk candidates score above it -- counting
        // all candidates would let the extras hide an underflow (clustered rows in a sample tile: 72 above, 100+ collected).
        float floor_mul[HDB_FUSED_MAXQ];             // thr_out is in the comparison domain: x this = score domain of the entries
        float qsq_fin[HDB_FUSED_MAXQ];               // euclidean: ||q||^2 for the re-score of near-duplicates
#pragma unroll
        for (int q = 0; q < HDB_FUSED_MAXQ; ++q) {
            floor_mul[q] = 1.f; qsq_fin[q] = 0.f;
            if (q < nq) {
                if (qpar[HDB_FUSED_MAXQ + q] != 0.f) qnan_bits |= 1u << q;
                if (!HAS_BIAS && METRIC != 2) floor_mul[q] = qpar[q];
                if (METRIC == 2) qsq_fin[q] = qpar[3 * HDB_FUSED_MAXQ + q];
            }
        }
        __syncthreads();                             // everyone has its copy: fbuf may now cover ctl / qpar
        // The status words are written LAST, behind a system-scope release: a host that polls them in a pinned record
        // (hdb_topk_host) may read the results as soon as every word has left its sentinel value.
        int32_t my_status = 0;
#pragma unroll 1
        for (int q = 0; q < nq; ++q) {
            const uint32_t tot0 = __hip_atomic_load(f.ctl + 2 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t tot = aborted ? 0u : tot0;
            uint32_t tot_eff = tot, ovf_eff = 0u;        // (slotted lists: the sum of the workgroups' counts; a workgroup with more rows than its slot holds = overflow)
            float fm = 1.f, ssq = 0.f;
#pragma unroll
            for (int qq = 0; qq < HDB_FUSED_MAXQ; ++qq) if (qq == q) { fm = floor_mul[qq]; ssq = qsq_fin[qq]; }
            // euclidean scores come from ||v||^2 + ||q||^2 - 2 v.q, which cancels when v ~ q: candidates closer than 5 % of
            // ||q||^2 are re-scored from the stored row with the direct difference (hdb_rescore_euclid_kernel, reference :49)
            auto rescore = [&](unsigned long long* buf, uint32_t nc) {
                if constexpr (METRIC == 2 && sizeof(E) == 2) {
                    const float* qv = f.Qraw + (int64_t)q * D;
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
            // local flavour: the floor is the highest b_w of any workgroup (score-domain keys in the granule area)
            // local flavour: the floor is the highest b_w of any workgroup (score-domain keys in the granule area, with the
            // workgroups' entry counts: the slotted lists have no counter of their own)
            uint32_t maxb = 0u, ltot = tot, lovf = 0u;
            uint32_t* wmb = reinterpret_cast<uint32_t*>(smem + (size_t)HDB_CAND_CAP * 16 + 2048 * 4 + 64);      // [0..7] wave maxima, [8..15] wave sums, [16] overflow, [32 ..] counts per workgroup
            if (local) {
                uint32_t bk = 0u, cs = 0u;
                if (tid == 0) wmb[16] = 0u;
                for (int64_t wg = tid; wg < G; wg += 512) {
                    const unsigned long long gw = __hip_atomic_load((hdb_gu64*)(f.gran + (wg * HDB_FUSED_GRAN_PER_WG + q)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t c = (uint32_t)(gw >> 32);
                    bk = max(bk, (uint32_t)gw);
                    cs += c < f.local_slot ? c : f.local_slot;
                    wmb[32 + wg] = c < f.local_slot ? c : f.local_slot;
                    if (SLOTTED && c > f.local_slot) lovf = 1u;
                }
                bk = hdb_wave_max_dpp(bk);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cs += (uint32_t)__shfl_xor((int)cs, o, 64);
                if (lane == 0) { wmb[w] = bk; wmb[8 + w] = cs; }
                __syncthreads();
                if (lovf) wmb[16] = 1u;
                uint32_t sum = 0u;
#pragma unroll
                for (int w2 = 0; w2 < 8; ++w2) { maxb = max(maxb, wmb[w2]); sum += wmb[8 + w2]; }
                if (SLOTTED) ltot = aborted ? 0u : sum;
                __syncthreads();
                lovf = wmb[16];
            }
            uint32_t kth_above;
            if constexpr (METRIC == 2 && sizeof(E) == 2)
                kth_above = hdb_finalize_fast(fbuf, f.cand + (int64_t)q * f.cap, tot, q, f.cap, f.k, f.kk, f.row_base, f.idx_out,
                                              f.score_out, nullptr, 0, 0, f.thr_out + q, fm, rescore, local, maxb);
            else
                kth_above = hdb_finalize_fast(fbuf, f.cand + (int64_t)q * f.cap, ltot, q, f.cap, f.k, f.kk, f.row_base, f.idx_out,
                                              f.score_out, nullptr, 0, 0, f.thr_out + q, fm, HdbNoFix(), local, maxb,
                                              local ? wmb + 32 : nullptr, f.local_slot, (uint32_t)G);
            if (local && SLOTTED) { tot_eff = ltot; ovf_eff = lovf; }
            if (tid == q) {
                const uint32_t nc = tot_eff < f.cap ? tot_eff : f.cap;
                my_status = ((tot_eff > f.cap || ovf_eff) ? HDB_Q_OVERFLOW : 0) | ((nc < f.kk || (f.kk > 0 && !kth_above)) ? HDB_Q_UNDERFLOW : 0) | (((qnan_bits >> q) & 1u) ? HDB_Q_NAN : 0);
            }
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every thread's result stores are out
        __syncthreads();
        if (tid < nq && f.status) __hip_atomic_store(f.status + tid, my_status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        HDB_STAMP(6);
        if (tid < 2 + HDB_FUSED_MAXQ) __hip_atomic_store(f.ctl + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next call
        if (tid == 32) __hip_atomic_store(f.ctl + 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static size_t fused_lds_bytes(int stage_bytes, size_t pend_bytes) {
    const size_t scan = mfma_lds_bytes(stage_bytes) + 2 * HDB_FUSED_MAXQ * 64 * 4 + 16 * 4 + pend_bytes + 64;
    const size_t fin = (size_t)HDB_CAND_CAP * 16 + 2048 * 4 + 64 + 128 + HDB_FUSED_MAX_WG * 4;      // hdb_finalize_body in the last workgroup (+ wave maxima / sums and the workgroups' counts of the local flavour)
    return scan > fin ? scan : fin;
}

template <typename E, int VQ, int D, int R, int METRIC, bool HAS_BIAS>
static int launch_fused_one(const ScanArgs& a, const FusedArgs& f, const float* aux0, int blocks, hipStream_t st) {
    auto kern = hdb_mfma_fused_kernel<E, VQ, D, R, METRIC, HAS_BIAS>;
    const size_t lds = fused_lds_bytes(R * D * (int)sizeof(E), sizeof(E) == 4 ? (size_t)2 * HDB_FUSED_PEND * R * 4 : 0);
    static unsigned long long attr_done = 0;
    hipError_t e = hdb_lds_attr_once(reinterpret_cast<const void*>(kern), (int)lds, &attr_done);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, st, a, f, aux0);
    return (int)hipGetLastError();
}

template <typename E, int VQ, int D, int R>
static int launch_fused(const ScanArgs& a, const FusedArgs& f, int blocks, hipStream_t st) {
    const bool bias = a.bias != nullptr;
    if (a.metric == HDB_DOT) return bias ? launch_fused_one<E, VQ, D, R, 0, true>(a, f, nullptr, blocks, st) : launch_fused_one<E, VQ, D, R, 0, false>(a, f, nullptr, blocks, st);
    if (a.metric == HDB_COSINE) return bias ? launch_fused_one<E, VQ, D, R, 1, true>(a, f, a.inv_norm, blocks, st) : launch_fused_one<E, VQ, D, R, 1, false>(a, f, a.inv_norm, blocks, st);
    if (a.metric == HDB_PEARSON) return bias ? launch_fused_one<E, VQ, D, R, 3, true>(a, f, a.inv_norm, blocks, st) : launch_fused_one<E, VQ, D, R, 3, false>(a, f, a.inv_norm, blocks, st);   // a.inv_norm carries 1/(sd_v d)
    if constexpr (sizeof(E) == 4 || D != 768)       // euclidean: a.inv_norm carries ||v||^2 (set by the host; unused by the float32 flavour); fp16 d = 768 spills 3 registers with it
        if (a.metric == HDB_EUCLIDEAN) return bias ? launch_fused_one<E, VQ, D, R, 2, true>(a, f, a.inv_norm, blocks, st) : launch_fused_one<E, VQ, D, R, 2, false>(a, f, a.inv_norm, blocks, st);
    return (int)hipErrorNotSupported;
}
