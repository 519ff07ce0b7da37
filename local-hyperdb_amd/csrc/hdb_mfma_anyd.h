// hdb_mfma_anyd.h -- the MFMA row scan for rows of ANY width that is a multiple of 16 bytes (fp16: d % 8 == 0, float32: d % 4 == 0):
// d = 96 / 200 / 300 / 1000 (GloVe, fastText, ...) instead of the multiples of 128 the geometries are cut for.  The reference takes
// any d (hyperdb/ranking_algorithm.py:29,:41); round 3 sent such matrices to the VALU scan, four queries per pass.
//
// No new kernel: the K-slice flavour of hdb_mfma_kernel (KSL) already reads rows at a run-time pitch.  A matrix of width d rides the
// geometry of the next instantiated width Dp >= d as ONE slice -- row pitch d * sizeof(E), ks_valid = the bytes of a row that exist.
// The staging fetches chunk 0 of the row again in place of the chunks past its end (no read outside the matrix, no extra HBM
// traffic: the line is in flight anyway) and the query fragments are zero there, so those products vanish.  V is read exactly once
// per pass; the matrix pipe multiplies Dp / d times the useful work, which an HBM-bound pass of up to 128 queries does not see.
// Multi-kernel pipeline only (sample scan, threshold, filter scan, finalize: MODE 0 / 1).
#pragma once
#include "hdb_mfma_kernel.h"

template <typename E, int D, int R, int KP = 1>
static int launch_anyd(const ScanArgs& a, int mode, const void* q, const float* sqnorm, const float* qsq, const float* qscl, int nq_launch, int blocks, hipStream_t st) {
    const bool b = a.bias != nullptr;
#define HDB_AD_CASE(MODE_)                                                                                                                        \
    if (a.metric == HDB_DOT) return b ? launch_kslice_one<E, D, R, MODE_, 0, true, KP>(a, q, nullptr, qsq, qscl, nq_launch, blocks, st)                \
                                      : launch_kslice_one<E, D, R, MODE_, 0, false, KP>(a, q, nullptr, qsq, qscl, nq_launch, blocks, st);              \
    if (a.metric == HDB_COSINE) return b ? launch_kslice_one<E, D, R, MODE_, 1, true, KP>(a, q, a.inv_norm, qsq, qscl, nq_launch, blocks, st)          \
                                         : launch_kslice_one<E, D, R, MODE_, 1, false, KP>(a, q, a.inv_norm, qsq, qscl, nq_launch, blocks, st);        \
    if (a.metric == HDB_EUCLIDEAN) return b ? launch_kslice_one<E, D, R, MODE_, 2, true, KP>(a, q, sqnorm, qsq, qscl, nq_launch, blocks, st)           \
                                            : launch_kslice_one<E, D, R, MODE_, 2, false, KP>(a, q, sqnorm, qsq, qscl, nq_launch, blocks, st);
    if (mode == 0) { HDB_AD_CASE(0) } else if (mode == 1) { HDB_AD_CASE(1) }
#undef HDB_AD_CASE
    return (int)hipErrorNotSupported;
}

// one slice over the whole (narrower) row
static inline ScanArgs anyd_args(const ScanArgs& in, int es) {
    ScanArgs a = in;
    a.ks_pitch = (int64_t)in.d * es; a.ks_off = 0; a.ks_dfull = in.d; a.ks_valid = in.d * es;
    a.ks_partial_in = nullptr; a.ks_partial_out = nullptr;
    return a;
}
