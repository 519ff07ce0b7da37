/* hyperdb_hip.h -- C ABI of the MI355X (gfx950) brute-force ranking engine.
 *
 * Drop-in boundary for the hot path of AdamCodd/local-hyperDB.  The reference has no FFI
 * layer: its boundary is the Python function table of hyperdb/ranking_algorithm.py, called
 * from hyperdb/hyperdb.py:1556.  Every entry point below cites the reference interface it
 * replaces; the Python side (local-hyperdb_amd/hyperdb/ranking_algorithm.py) keeps the
 * reference's names and signatures and reaches these symbols through ctypes.
 *
 * Conventions
 *   - All pointers named dev_* are device (HBM) pointers owned by the caller (PyTorch-ROCm
 *     tensors on the Python side).  The library BORROWS them; it owns only its workspace.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only
 *     enqueue work; they do not synchronise unless stated.
 *   - Return value: 0 (HDB_OK) or a negative hdb_status; the message of the last failure on
 *     the calling thread is returned by hdb_last_error().
 *   - One in-flight call per handle (the reference is single-threaded, hyperdb.py:1381-1388).
 *   - Scores leave the device as float32; the Python shim widens to float64 like
 *     ranking_algorithm.py:171.  Indices are int64 like numpy's argpartition output (:199).
 */
#ifndef HYPERDB_HIP_H
#define HYPERDB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hdb_index hdb_index;

/* dtype of the stored matrix: HyperDB(fp_precision=...) accepts float16/32/64 (hyperdb.py:65-66). */
enum hdb_dtype { HDB_F16 = 0, HDB_F32 = 1, HDB_F64 = 2 };

/* metric strings of hyperDB_ranking_algorithm_sort's dispatch table (ranking_algorithm.py:155-163). */
enum hdb_metric {
    HDB_DOT = 0,        /* dot_product          ranking_algorithm.py:24-30   */
    HDB_COSINE = 1,     /* cosine_similarity    ranking_algorithm.py:32-42   */
    HDB_EUCLIDEAN = 2,  /* euclidean_metric     ranking_algorithm.py:44-52   (similarity 1/(1+dist)) */
    HDB_HAMMING = 3,    /* hamming_distance     ranking_algorithm.py:128-147 (d - popcount(xor of x>0)) */
    HDB_MANHATTAN = 4,  /* manhattan_distance   ranking_algorithm.py:54-61   */
    HDB_JACCARD = 5,    /* jaccard_similarity   ranking_algorithm.py:63-75   */
    HDB_PEARSON = 6,    /* pearson_correlation  ranking_algorithm.py:77-113  */
    HDB_EUCLIDEAN_DIST = 7 /* euclidean_metric(get_similarity_score=False): raw distance, hdb_scores only */
};

enum hdb_status {
    HDB_OK = 0,
    HDB_ERR_ARG = -1,          /* bad shape / dtype / k / null pointer  -> ValueError in the shim  */
    HDB_ERR_HIP = -2,          /* a HIP runtime call failed             -> RuntimeError            */
    HDB_ERR_UNSUPPORTED = -3,  /* metric/dtype combination not built    -> NotImplementedError     */
    HDB_ERR_NOMEM = -4
};

/* per-query status bits written by hdb_topk (device int32 per query) */
enum hdb_query_status {
    HDB_Q_OK = 0,
    HDB_Q_UNDERFLOW = 1,  /* sampled threshold too high: fewer than k candidates passed */
    HDB_Q_OVERFLOW = 2,   /* candidate buffer overflowed (massive ties or skewed sample) */
    HDB_Q_NAN = 4         /* the query vector contains a NaN (ValueError of ranking_algorithm.py:150-151) */
};

/* Library/ABI version (major*100+minor). */
int hdb_version(void);

/* Message of the last error on this thread ("" if none). */
const char* hdb_last_error(void);

/* Register a resident N x d row-major matrix (C-contiguous, like HyperDB.vectors, hyperdb.py:80,:911).
 * One pass over V builds the per-row caches that the reference recomputes on every query:
 * 1/||v|| (get_norm_vector, ranking_algorithm.py:8-21, zero norm -> 1), ||v||^2, and the NaN
 * flag (the np.isnan(vectors).any() of ranking_algorithm.py:150).  `row_base` is added to every
 * returned index (global row id of local row 0 when the matrix is one shard of a larger one).
 * Enqueues on `stream`; the index is usable on the same stream immediately. */
int hdb_index_create(hdb_index** out, const void* dev_V, int64_t n, int32_t d, int dtype,
                     int device, int64_t row_base, void* stream);

/* Re-point an index at a (possibly grown / rewritten) matrix and rebuild the row caches;
 * matrix lifecycle counterpart of commit_pending / remove_document (hyperdb.py:503-509,:721-728). */
int hdb_index_update(hdb_index* ix, const void* dev_V, int64_t n, void* stream);

/* Growable matrix (HyperDB.add, hyperdb.py:503-509 grows self.vectors by np.concatenate on every commit):
 * hdb_index_rebase -- the caller moved the SAME rows to another allocation (capacity doubling); only the borrowed
 *                     pointer changes, every cache stays valid.
 * hdb_index_extend -- rows [n_old, new_n) were appended behind the existing rows of the current allocation; the
 *                     1/||v||, ||v||^2 and NaN caches are extended over the new rows only (O(new rows), not O(N));
 *                     the sign bits and the pearson row scales are extended the same way on their next use (lazy caches:
 *                     the appended rows only). */
int hdb_index_rebase(hdb_index* ix, const void* dev_V);
int hdb_index_extend(hdb_index* ix, int64_t new_n, void* stream);

/* Global row id of local row 0 (see hdb_index_create): a shard's base moves when an earlier shard grows or is compacted. */
int hdb_index_set_row_base(hdb_index* ix, int64_t row_base);

/* Compaction after HyperDB.remove_document (hyperdb.py:691-766; the reference rebuilds self.vectors on the host with
 * np.vstack / a boolean mask, :721-728): the m kept rows dev_rows[0..m) (ascending local row ids, int64, device) are
 * gathered into dev_V_out (m x d, caller-owned, must not alias the current matrix) in one pass at HBM speed, and the
 * 1/||v||, ||v||^2 and NaN caches travel with their rows -- nothing is recomputed.  On return (the call synchronises
 * `stream`) the index borrows dev_V_out with n = m and the caller may release the old matrix; bias and mask are
 * cleared, sign-bit / pearson caches are rebuilt lazily. */
int hdb_index_gather(hdb_index* ix, const int64_t* dev_rows, int64_t m, void* dev_V_out, void* stream);

void hdb_index_destroy(hdb_index* ix);

/* 1 if the matrix contains a NaN (synchronises `stream` of the create/update call). */
int hdb_index_has_nan(hdb_index* ix, int* out_flag);

/* Additive per-row term applied before top-k: the recency_scores of ranking_algorithm.py:180-186.
 * dev_bias: n floats or NULL to clear.  Borrowed until replaced/cleared. */
int hdb_index_set_bias(hdb_index* ix, const float* dev_bias);

/* Device-side builder of that term: dev_out[i] = recency_bias * exp(dev_ts[i] - ts_max), evaluated in
 * float64 and stored as float32 (ranking_algorithm.py:183; ts_max = max(timestamps), which the caller
 * already knows from the host-side timestamp list, hyperdb.py:1341-1344). */
int hdb_recency_bias(const double* dev_ts, int64_t n, double recency_bias, double ts_max, float* dev_out,
                     int device, void* stream);

/* Both decays a HyperDB.query() call applies (hyperdb.py:1344 over the FILTERED documents, then ranking_algorithm.py:183 on
 * those values): dev_out[i] = rb * exp(first_i - max first), first_i = rb * exp(-ts_max + dev_ts[i]), for the rows dev_mask keeps
 * (NULL: all rows; dropped rows get 0).  ts_max / ts_min = newest / oldest timestamp among the kept rows (the maximum of
 * `first` sits at one of them).  float64 arithmetic, float32 result: the whole recency term of the facade without a host pass. */
int hdb_recency_bias_twice(const double* dev_ts, const uint8_t* dev_mask, int64_t n, double recency_bias, double ts_max,
                           double ts_min, float* dev_out, int device, void* stream);

/* Optional row subset (filters / skip_doc, hyperdb.py:1119-1134,:1258-1308): dev_mask is n bytes,
 * non-zero = row takes part; NULL clears.  Excluded rows score -inf and are never returned
 * while at least k rows are included. */
int hdb_index_set_row_mask(hdb_index* ix, const uint8_t* dev_mask);

/* Full score vector of one query: the per-metric functions dot_product / cosine_similarity /
 * euclidean_metric / hamming_distance (... :24,:32,:44,:128).  dev_q: d elements, float32 for
 * F16/F32 matrices, float64 for F64 matrices.  dev_out: n floats.  The bias is NOT added. */
int hdb_scores(hdb_index* ix, const void* dev_q, int metric, float* dev_out, void* stream);

/* Top-k of nq independent queries: metric scoring + NaN->-inf + bias + argpartition/argsort of
 * hyperDB_ranking_algorithm_sort (ranking_algorithm.py:168-204), for nq queries at once (the
 * reference takes one query per call).  dev_Q: nq x d row-major (float32, or float64 for F64
 * matrices).  Outputs: dev_idx [nq][k] int64 (row_base added; -1 where fewer than k rows exist),
 * dev_score [nq][k] float32, sorted by (score descending, index ascending).
 * dev_status [nq] int32 receives hdb_query_status bits; queries with a non-zero status must be
 * re-run with hdb_topk_exact (the threshold estimate from the row sample failed for them).
 * On fp16 matrices (d any multiple of 128 up to 1536; dot, cosine, euclidean, pearson) the
 * scores come from the matrix cores with fp16 copies of the queries (scaled per query by a power of two, so any
 * float32 magnitude is safe) and float32 accumulation: nothing is lost when the query has the matrix's dtype, a
 * float32 query is rounded to 11 significant bits per element (score error ~1e-4 relative, inside the 1e-3
 * contract for fp16 data).  hdb_set_option(ix, "use_mfma", 0) keeps float32 queries unrounded (VALU scan).
 * float32 matrices (d in {128,256,384,512,768}) take batches of 5+ queries through fp32 MFMAs: exact fp32 products.
 * Calls of 1-4 dot / cosine queries with k <= 128 on an fp16 matrix (d = 256 .. 768; 1-2 queries for d = 1024 .. 1536), or of 1-2 on a float32 matrix
 * (d in {128,256,384}; one query at d = 768 and, from 1.5 M rows on, at d = 512), run as ONE kernel launch (query preparation, row sample, threshold exchange between the
 * workgroups, filter pass, final sort: hdb_mfma_fused.h); everything else is the same pipeline as separate launches.
 * Results are bit-identical either way. */
int hdb_topk(hdb_index* ix, const void* dev_Q, int32_t nq, int32_t k, int metric,
             int64_t* dev_idx, float* dev_score, int32_t* dev_status, void* stream);

/* hdb_topk + copy of the packed result record ([nq*k int64][nq*k f32][nq i32], hdb_packed_bytes) into host memory
 * (pinned memory recommended) + stream synchronisation, in one call: what one HyperDB.query() needs.  Queries whose
 * sampled threshold failed are re-run through the exact path before returning; on return every status word is 0
 * except HDB_Q_NAN.  Saves the caller-side allocations and the extra host round trips of doing this in Python. */
int hdb_topk_host(hdb_index* ix, const void* dev_Q, int32_t nq, int32_t k, int metric, void* host_record, void* stream);

/* Same contract, but by materialising all n scores per query and radix-selecting them:
 * always exact, any tie pattern, any k <= HDB_MAX_K; dev_status is written as 0. */
int hdb_topk_exact(hdb_index* ix, const void* dev_Q, int32_t nq, int32_t k, int metric,
                   int64_t* dev_idx, float* dev_score, int32_t* dev_status, void* stream);

/* Merge `parts` per-shard top-k lists (the all-gathered [parts][nq][k] buffers of the row-sharded
 * index) into the global top-k per query, same ordering rule.  Every part list must be ordered as hdb_topk writes
 * it: score descending, row ascending, unused slots (index -1) at the end. */
int hdb_merge_topk(const int64_t* dev_idx_parts, const float* dev_score_parts, int32_t parts,
                   int32_t nq, int32_t k, int64_t* dev_idx, float* dev_score, int device, void* stream);

/* Packed per-shard result record used for the exchange step (ONE RCCL all-gather per query batch):
 *   [nq*k int64 indices][nq*k float32 scores][nq int32 status], padded to a multiple of 16 bytes.
 * hdb_packed_bytes gives the record size; hdb_topk can write straight into such a record (pass
 * base, base + nq*k*8, base + nq*k*12).  hdb_merge_topk_packed merges `parts` gathered records and
 * ORs their status words per query into dev_status (may be NULL). */
int64_t hdb_packed_bytes(int32_t nq, int32_t k);
int hdb_merge_topk_packed(const void* dev_gathered, int32_t parts, int32_t nq, int32_t k, int64_t* dev_idx,
                          float* dev_score, int32_t* dev_status, int device, void* stream);
/* The same merge on the HOST, for records that are already in host memory (one process per GPU on one node: every rank's
 * hdb_topk_host leaves its record in host memory, the ranks swap the 1.2 KB records through a shared-memory segment and
 * merge here -- a few microseconds instead of a collective, a merge launch and another synchronisation).  `records` =
 * `parts` packed records back to back (shard p's rows precede shard p+1's), `out_record` = one packed record; same total
 * order (score descending, global row ascending), status words OR-ed per query, missing entries -1 / -inf.  No GPU call. */
int hdb_merge_topk_host(const void* records, int32_t parts, int32_t nq, int32_t k, void* out_record);
/* The swap itself, for `world` processes of one node that have all mapped the same zero-initialised shared-memory segment
 * `shm` of 2 * world * stride bytes (hyperdb/sharded.py HostExchange creates it): slot (parity, rank) = 64-byte header
 * {seq} + record.  Exchange number seq (1, 2, ... in lockstep on every rank) copies `record` (hdb_packed_bytes(nq, k)
 * bytes, <= stride - 64) into this rank's slot of parity seq & 1, publishes seq behind it (release), waits until every
 * rank's slot carries seq (acquire; HDB_ERR_HIP after timeout_s) and merges the `world` records straight out of the
 * segment into out_record like hdb_merge_topk_host.  A slot is rewritten two exchanges later, which no rank can reach
 * before everybody has published the exchange in between, i.e. has finished reading this one.  No GPU call. */
int hdb_host_exchange_merge(void* shm, int64_t stride, int32_t world, int32_t rank, uint64_t seq, const void* record,
                            int32_t nq, int32_t k, void* out_record, double timeout_s);

/* Single-process multi-GPU group: HyperDB.query() (hyperdb.py:1584) is a single-process
 * call, so the row-sharded matrix must be reachable without a launcher.  A group is `parts` row shards, each an hdb_index
 * created on its own device with its row_base (a device may hold several shards).  hdb_group_topk_host copies the
 * queries (HOST memory, nq x d float32, or float64 for F64 matrices) into a pinned staging buffer every device can read,
 * runs hdb_topk_host on every shard concurrently (one worker thread and one stream per shard; between calls a worker spins
 * briefly on the job word, then parks), lets each shard's last kernel store its packed record into a pinned, portable
 * host buffer that every device can write, merges the `parts` records on the host (as hdb_merge_topk_host) and returns
 * the merged packed record (hdb_packed_bytes layout) in host_record.  A shard whose sampled threshold failed re-runs those
 * queries locally through the exact selection before it reports: on return every status word is 0 except HDB_Q_NAN.
 * Bias / mask are set per shard on the shards' own handles (recency: pass the GLOBAL newest timestamp as ts_max to
 * hdb_recency_bias).  Any k the single index takes is accepted (the merge runs on the host).
 * The group borrows the shard handles: destroy the group first, then the shards. */
typedef struct hdb_group hdb_group;
int hdb_group_create(hdb_group** out, hdb_index* const* shards, int32_t parts);
int hdb_group_topk_host(hdb_group* g, const void* host_Q, int32_t nq, int32_t k, int metric, void* host_record);
void hdb_group_destroy(hdb_group* g);

/* Largest k served by the selection kernels.  hdb_topk / hdb_topk_exact accept any k: above HDB_MAX_K (on a
 * matrix of more than 8192 rows) they materialise the scores of each query and radix-sort them (cold path,
 * "top_k > N returns all rows sorted", ranking_algorithm.py:195-200).
 * The two DEVICE merges, hdb_merge_topk and hdb_merge_topk_packed, rank parts*k entries per query in LDS and return
 * HDB_ERR_UNSUPPORTED beyond parts*k = 8192 (8 ranks: k > 1024).  Nothing above the C ABI is limited by that:
 * hdb_merge_topk_host, hdb_host_exchange_merge and hdb_group_topk_host merge on the host with any parts*k, and the
 * one-process-per-GPU path (hyperdb/sharded.py) copies the all-gathered records to the host and calls
 * hdb_merge_topk_host whenever parts*k exceeds the cap, with either exchange transport. */
#define HDB_MAX_K 2048

/* Tuning knobs / introspection (bench and tests): name -> value, returns HDB_ERR_ARG if unknown.
 * Options:  max_blocks (0 = automatic grid of the row scans), force_exact (1: always the exact selection),
 *   sample_target (expected survivors of the sampled threshold, 0 = automatic), use_mfma (0: VALU scans only),
 *   mfma_min_q (smallest batch that takes the MFMA scan on fp16 matrices, default 1), mfma_variant (16 | 32: MFMA
 *   shape of the 256-query pass), bits_fused (0: hamming / jaccard always through the exact selection; 3: one-query calls on 1M+ rows keep the six launches),
 *   host_direct (0: hdb_topk_host always copies through a device record), exact_bytes (score workspace cap of
 *   the exact path), finalize_threads (256 | 512 | 1024), profile (1: HIP events around the pass over V),
 *   use_fused (0: never the single-launch pipeline), fused_timeout_us (bound of its in-kernel spins, default 2000),
 *   host_poll (0: hdb_topk_host always waits on the stream instead of polling the status words of a pinned record),
 *   dyn_tiles (0: the batched MFMA filter pass always splits its tiles statically; default 1 = tiles from a counter for
 *   long passes over rows of >= 768 bytes with up to 64 queries).
 *   Dispatch of few-query calls (-1 = the measured rule): fused_max_q (the 1-4-query single launch takes
 *   calls of up to this many queries), f32_min_q (float32 matrices: the matrix-core scan from this many queries on), bits_max_q
 *   (hamming / jaccard: single launches of four queries up to this many queries, more through the six launches in one go);
 *   use_batch1 (0: never the batched single launch), use_l1_tile (0: manhattan batches stay with the 4-query scan).
 *   max_blocks < 0 asks for -max_blocks workgroups per CU in the batched MFMA scan (measured: no gain).
 * Stats:    path (0 small, 1 sampled threshold, 2 exact, 3 full sort), mfma, fused (0 multi-kernel, 1 the 1-4-query single launch,
 *   2 the batched single launch, 3 the bit-metric single launch), host_direct, chunks, sample_rows, sample_m,
 *   scan_launches, scan_time_ns (sum over the profiled launches), cand_cap, n, ws_bytes. */
int hdb_set_option(hdb_index* ix, const char* name, int64_t value);
int hdb_get_stat(hdb_index* ix, const char* name, int64_t* value);

#ifdef __cplusplus
}
#endif
#endif /* HYPERDB_HIP_H */
