// hdb_mfma_ksplit.hip -- the MFMA row scan for rows too wide for one wave's query fragments: float32 d = 1024 / 1536 (the
// reference's default precision at its demo width, hyperdb.py:51) and fp16 d = 2048 / 3072 / 4096.
//
// A wave holds the B fragments of 16 queries for at most 3072 bytes of row (192 registers), and a 16-row LDS stage of wider rows
// does not fit the 3-deep ring.  So the K dimension is cut into S slices that the EXISTING slice geometries cover (float32 512 /
// 768 elements, fp16 1024 / 1536): one launch per slice reads its piece of every row (row pitch = the full row), starts its
// accumulators from the sums of the slices before (a [query][rows] float32 buffer, the layout of MODE 0's scores) and either
// stores the raw sums for the next slice (MODE 3) or, in the last slice, runs the metric's epilogue as usual (scores or
// threshold filter).  V is still read exactly once per pass; the partial sums add 8 B per row, query and extra slice (Q = 64,
// float32 d = 1536: +8 % traffic).  Batches of 5-128 queries per pass instead of the VALU scan's 4 (a 256-query batch on a
// float32 d = 1536 matrix: 2 x 2 launches instead of 64 passes).
#include "hdb_mfma_kernel.h"

struct KsGeom { int slices; int dslice; };
static KsGeom ks_geom(int dtype, int d) {
    if (dtype == HDB_F32 && d == 1024) return {2, 512};
    if (dtype == HDB_F32 && d == 1536) return {2, 768};
    if (dtype == HDB_F16 && d == 2048) return {2, 1024};
    if (dtype == HDB_F16 && d == 3072) return {2, 1536};
    if (dtype == HDB_F16 && d == 4096) return {4, 1024};
    return {0, 0};
}
extern "C" int hdb_mfma_ksplit_slices(int dtype, int d) { return ks_geom(dtype, d).slices; }
extern "C" int hdb_launch_mfma_kslice_f32s(const ScanArgs* a, int dslice, int mode, int nq_launch, const void* q, const float* sqnorm,
                                           const float* qsq, int blocks, void* stream);      // hdb_mfma_ksplit_s.hip

// args->ks_partial_out: [nq_launch][ks_ld] float32 scratch of the caller (MODE 0 passes may alias it with args->scores: the last
// slice overwrites the sums with the scores, element by element, by the lane that read them)
extern "C" int hdb_launch_mfma_ksplit(const ScanArgs* args, int dtype, int mode, int nq_launch, const void* q, const float* sqnorm,
                                      const float* qsq, const float* qscl, int blocks, void* stream) {
    const KsGeom g = ks_geom(dtype, args->d);
    if (!g.slices || !args->ks_partial_out) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int es = dtype == HDB_F16 ? 2 : 4;
    for (int s = 0; s < g.slices; ++s) {
        ScanArgs a = *args;
        a.ks_pitch = (int64_t)args->d * es; a.ks_off = s * g.dslice * es; a.ks_dfull = args->d;
        a.ks_partial_in = s == 0 ? nullptr : args->ks_partial_out;
        const int m = s + 1 < g.slices ? 3 : mode;
        int rc;
        if (dtype == HDB_F32 && args->f32_split) rc = hdb_launch_mfma_kslice_f32s(&a, g.dslice, m, nq_launch, q, sqnorm, qsq, blocks, stream);
        else if (dtype == HDB_F32) rc = g.dslice == 512 ? launch_kslice<float, 512, 16>(a, m, q, sqnorm, qsq, nullptr, nq_launch, blocks, st)
                                                   : launch_kslice<float, 768, 16>(a, m, q, sqnorm, qsq, nullptr, nq_launch, blocks, st);
        else rc = g.dslice == 1024 ? launch_kslice<_Float16, 1024, 16>(a, m, q, sqnorm, qsq, qscl, nq_launch, blocks, st)
                                   : launch_kslice<_Float16, 1536, 16>(a, m, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
        if (rc) return rc;
    }
    return 0;
}
