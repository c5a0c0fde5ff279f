/*
 * rarc.h — C-ABI of librarc_hip.so, the MI355X (gfx950) backend for RAG-ARC's
 * dense-retrieval hot path.
 *
 * The reference (DataArcTech/RAG-ARC) has no FFI of its own: the hot path is
 * Python that calls faiss / numpy.  Each entry point below replaces the
 * arithmetic behind one reference call site (cited as file:line relative to the
 * reference tree).  INTEGRATION.md shows the ctypes stubs a maintainer adds on
 * the reference side.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HBM); h_* is a host pointer.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - every function returns 0 on success, a negative RARC_E_* code otherwise;
 *     nothing throws across the boundary; rarc_last_error() returns a
 *     thread-local human-readable message for the last failure.
 *   - the library allocates nothing: the caller owns all buffers, including the
 *     scratch workspace whose size rarc_search_workspace_bytes() reports.
 *   - functions are re-entrant per (device, stream); the caller selects the
 *     device (hipSetDevice) before calling.
 *   - doc ids are row indices (int64); scores are fp32.
 *   - ordering everywhere: score descending, ties by id ascending.
 */
#ifndef RARC_H
#define RARC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RARC_VERSION 600 /* 0.6.0: the encoder's QUERY PATH (RarcEnc32Layer.*_wq, rarc_enc32_pack_query_weight: forwards of 32 / 64 / 128 tokens as weight streams), rarc_stream_read (the measured HBM read ceiling a bench line reports next to the nominal one); 0.5.1: rarc_similar_pairs (all-pairs cosine >= threshold: the graph store's entity dedup); 0.5.0: rarc_search_wide (rows to 4096 padded dims, k to 8192), rarc_compact_rows (delete by compaction), rarc_vmem_* (growable arenas), RARC_E_IO / RARC_IO_TRUNCATE; 0.4.1: RarcEnc32Layer.f1_colmax (FFN1 with its GELU fused into the GEMM epilogue); 0.4.0: shard-file streaming (rarc_file_to_device, rarc_device_to_file), host WordPiece (rarc_wordpiece_*); 0.3.2: RMSNorm folded into the reranker LM's projections (RarcLmLayer.qkv_w_folded, gate_up_w_folded),
                            * rarc_enc_gemm_zero_bias; 0.3.1: relative-position attention bias in both encoder forwards (MPNet family: RarcEncModel / RarcEnc32Model
                            * rel_bias, rel_span); 0.3.0: fp32-class encoder forward (rarc_enc32_*) */

#define RARC_OK 0
#define RARC_E_INVALID -1     /* bad argument (null pointer, unsupported d/k, ...) */
#define RARC_E_HIP -2         /* a HIP runtime call or kernel launch failed */
#define RARC_E_WORKSPACE -3   /* workspace too small / misaligned */
#define RARC_E_UNSUPPORTED -4 /* shape outside what the kernels were built for */
#define RARC_E_IO -5          /* a file operation failed (open, read, write, fsync: ENOSPC, EIO, a short file ...) */

/* Limits of the fused scan kernels (int8 prefilter: d_pad <= 1024; fp16 scan: d_pad <= 768). */
#define RARC_MAX_QUERIES 256 /* queries per scan pass (register-resident) */
#define RARC_MAX_K 1024      /* largest k' the finalize kernel selects */
#define RARC_DIM_ALIGN 128   /* stored row length is a multiple of this */

/* status bits written per query by rarc_search_f16 (d_status[q]) */
#define RARC_Q_OK 0u
#define RARC_Q_UNCERTAIN 1u /* exactness certificate failed (fp16 scan only): call rarc_repair_f16 */
#define RARC_Q_OVERFLOW 2u  /* candidate buffer overflowed: call rarc_repair_f16 */
/* diagnostics in the second byte of a flagged query's word (which limit it ran into; hosts treat any non-zero word alike) */
#define RARC_Q_WHY_SEGMENT 0x100u    /* a scan workgroup's candidate segment for this query filled up */
#define RARC_Q_WHY_G1 0x200u         /* more rows at or above the k-th approximate score than the rescore buffer holds, no cut found */
#define RARC_Q_WHY_G2 0x400u         /* rows within the error bound of the k-th exact score exceed the buffer (unbanded path) */
#define RARC_Q_WHY_BAND_TIES 0x800u  /* banded rescore: a band of equal approximate scores does not fit */
#define RARC_Q_WHY_BAND_GUARD 0x1000u /* banded rescore: iteration guard exhausted */

int rarc_version(void);
const char* rarc_last_error(void);

/* Smallest multiple of RARC_DIM_ALIGN that is >= d. */
int rarc_padded_dim(int d);

/*
 * L2-normalise rows in fp32, in place or out of place (d_out may equal d_in).
 * Replaces faiss.normalize_L2 at
 *   encapsulation/database/vector_db/VectorStore_Faiss.py:150-154 (called from :178, :259).
 * Semantics (restated from faiss fvec_renorm_L2): nr = sum x^2 in fp32;
 * if nr > 0: x *= (float)(1.0 / sqrt(nr)); zero rows are left unchanged.
 * The summation order is the canonical 8-lane order documented in DESIGN.md.
 * ld_in / ld_out are row strides in elements.
 */
int rarc_l2norm_rows_f32(const float* d_in, int64_t ld_in, float* d_out, int64_t ld_out,
                         int64_t n_rows, int d, void* stream);

/*
 * Cosine similarities in float64, as the chunker computes them (numpy branch of
 *   core/file_management/chunker/spliter.py:307-332  cosine_similarity(X, Y)
 * = np.dot(X, Y.T) / np.outer(|X|, |Y|), entries that come out NaN/inf (a zero row) set to 0), and the distances of
 * consecutive rows  1 - cosine(x_i, x_{i+1})  of
 *   core/file_management/chunker/spliter.py:354-371  calculate_cosine_distances.
 * Inputs are fp32 rows on the device (what an Embeddings provider returns), sums run in fp64: products of fp32
 * values are exact there, so only the order of the additions differs from numpy (|delta| ~ 1e-16 relative).
 * d_out: [nx][ny] doubles / [n_rows - 1] doubles.
 */
int rarc_cosine_matrix_f32(const float* d_x, int64_t ld_x, int nx, const float* d_y, int64_t ld_y, int ny, int d,
                           double* d_out, void* stream);
int rarc_adjacent_cosine_distance_f32(const float* d_x, int64_t ld_x, int n_rows, int d, double* d_out, void* stream);

/*
 * Ingest: fp32 rows -> (optionally L2-normalised) -> fp16 corpus rows of length
 * d_pad (zero padded).  Replaces `_normalize_vectors` + `index.add` at
 *   VectorStore_Faiss.py:178, :199-202
 * for an HBM-resident fp16 flat index.  d_row_norm2 (optional, n_rows floats)
 * receives the squared L2 norm of each stored (fp16-rounded) row.
 */
int rarc_ingest_f16(const float* d_in, int64_t ld_in, uint16_t* d_corpus_f16, int d_pad,
                    float* d_row_norm2, int64_t n_rows, int d, int normalize, void* stream);

/*
 * Quantisation metadata of an fp16 corpus shard, needed by the int8-prefilter scan of
 * rarc_search_f16 (no reference counterpart: it is part of `index.add`,
 * VectorStore_Faiss.py:199-202, for this backend).  d_qmeta is a caller-owned float array of
 * rarc_quant_meta_floats(n_rows) elements, ZEROED by the caller before the first call:
 *   [0]            R = max over stored rows of ||d - d8/s||_2 (the int8 image's residual norm);
 *                  only ever raised, so it stays valid when rows are appended
 *   [1]            rho: a bound of ||d - d_scanned|| over all rows when the scan reads a ROUNDED image of the
 *                  stored rows (fp32 storage, rarc_search_f32: the caller sets it); 0 otherwise
 *   [2..3]         reserved
 *   [4+2t], [5+2t] scale s_t of the 32-row tile t (127 / max|x| rounded down to fp16) and 1/s_t
 * The call (re)computes the entries of every tile that intersects rows [first_row, n_rows): after
 * appending rows, pass the old row count as first_row.  The corpus buffer must hold ceil32(n_rows)
 * rows (the rows beyond n_rows may hold anything finite).
 */
size_t rarc_quant_meta_floats(int64_t n_rows);
int rarc_quant_meta_f16(const uint16_t* d_corpus_f16, int64_t n_rows, int d_pad, int64_t first_row,
                        float* d_qmeta, void* stream);

/*
 * Test hook: the dense matrix of int8-prefilter scores of a small shard (n_rows <= 2^22),
 * d_out [nq][n_rows] fp32, computed with the scan's own quantisation routine.  Lets a test check
 * |canonical - approx| <= eps8[q] for every (query, row) pair; not used by any search.
 */
int rarc_debug_q8_scores(const uint16_t* d_corpus_f16, int64_t n_rows, int d_pad, const float* d_qmeta,
                         const void* d_qblock, int nq, float* d_out, void* stream);

/*
 * Query preparation: fp32 queries [nq][ld_in] -> the "query block" the search reads:
 * (optionally L2-normalised) fp32 copy [RARC_MAX_QUERIES][d_pad] (zero padded), fp16 copy (seed
 * pass / fp16 scan), int8 copy + scale (int8 prefilter), and the per-query error bounds of both
 * approximate scorers.  Replaces `np.array([embedding]).astype(np.float32)` +
 * `_normalize_vectors` at VectorStore_Faiss.py:258-259.
 *   d_qmeta  : the corpus' quantisation metadata (its R enters the int8 error bound); NULL when
 *              the search will be run without it
 *   d_qblock : rarc_query_block_bytes(d_pad) bytes, 256-byte aligned, caller-owned
 */
size_t rarc_query_block_bytes(int d_pad);
int rarc_prep_queries(const float* d_in, int64_t ld_in, int nq, int d, int d_pad, int normalize,
                      float corpus_max_norm, const float* d_qmeta, void* d_qblock, void* stream);

/*
 * Re-running a search whose first attempt was flagged (RARC_Q_OVERFLOW / RARC_Q_UNCERTAIN): the k-th entry of the
 * incomplete answer is the canonical score of a real row, hence a lower bound of the true k-th best score.  After
 * rarc_prep_queries of the flagged queries, rarc_qblock_set_floor stores that bound per query (d_prev_ids /
 * d_prev_scores: [nq][k], the rows of the first answer that belong to these queries; entries with id < 0 give no
 * bound) and the next rarc_search_* starts from it instead of from a sample statistic: the candidate lists hold
 * only rows within one error bound of the final threshold.  rarc_prep_queries resets the bounds to -inf.
 */
int rarc_qblock_set_floor(void* d_qblock, int d_pad, const int64_t* d_prev_ids, const float* d_prev_scores, int k,
                          int nq, void* stream);

/* Bytes of device scratch rarc_search_f16 / rarc_repair_f16 need (256-byte aligned base).
 * After a search, the uint32 at byte offset RARC_WS_ANYFLAG_OFFSET of the workspace is the OR of
 * all d_status words of that search (0 == nothing to repair). */
#define RARC_WS_ANYFLAG_OFFSET 5124
size_t rarc_search_workspace_bytes(int cand_cap);

/*
 * Flat inner-product search over an fp16 corpus shard resident in HBM.
 * Replaces faiss.IndexFlatIP.search at VectorStore_Faiss.py:262-263 for
 * nq <= RARC_MAX_QUERIES queries at once (the reference calls it with nq = 1).
 *
 *   d_corpus_f16 : [ceil32(n_rows)][d_pad] fp16, row-major
 *   d_qmeta      : quantisation metadata (rarc_quant_meta_f16).  Non-NULL selects the int8
 *                  prefilter scan: rows are discarded on int8 MFMA scores under a rigorous error
 *                  bound, every possible top-k row is rescored canonically (exact by construction).
 *                  NULL selects the fp16 MFMA scan with k' selection + exactness certificate
 *                  (d_pad <= 768 only; kept for comparison).
 *   d_qblock     : output of rarc_prep_queries for these queries (same d_pad, same d_qmeta)
 *   k            : results per query (k <= kprime)
 *   kprime       : rows the pruning thresholds are built on (k <= kprime <= RARC_MAX_K); the fp16
 *                  path also rescored exactly this many
 *   id_base      : added to local row indices (global id of this shard's row 0)
 *   d_out_ids    : [nq][k] int64, -1 where fewer than k rows exist
 *   d_out_scores : [nq][k] fp32 canonical scores, -inf padding
 *   d_status     : [RARC_MAX_QUERIES + 1] uint32, ZEROED BY THE CALLER: entry q < nq receives
 *                  the RARC_Q_* bits of query q (a query whose bit is set must be repaired
 *                  with rarc_repair_f16 before its row is trusted); the last entry
 *                  receives the OR of all of them (one word to read back per batch)
 *   bin_lo/bin_hi: score range covered by the pruning histogram
 *                  (cosine: -1, +1; ip: -/+ max|q|*max|d|)
 */
int rarc_search_f16(const uint16_t* d_corpus_f16, int64_t n_rows, int d_pad, const float* d_qmeta,
                    const void* d_qblock, int nq, int k, int kprime, int64_t id_base, float bin_lo,
                    float bin_hi, int64_t* d_out_ids, float* d_out_scores, uint32_t* d_status,
                    void* d_workspace, size_t workspace_bytes, int cand_cap, void* stream);

/*
 * Exact repair / verification of ONE query row: scans the whole shard with the
 * canonical fp32 scorer, collects every row that beats the current k-th result
 * and re-sorts.  After it returns the row is exact regardless of d_status.
 * d_found (1 uint32) receives the number of rows that beat the previous k-th
 * entry (0 == the previous answer was already exact).
 */
int rarc_repair_f16(const uint16_t* d_corpus_f16, int64_t n_rows, int d_pad, const void* d_qblock,
                    int q, int k, int64_t id_base, int64_t* d_out_ids, float* d_out_scores,
                    uint32_t* d_found, void* d_workspace, size_t workspace_bytes, void* stream);

/*
 * Batched exact verification: for queries q_first .. q_first + nq - 1 (nq <= 8) of the prepared query block, count the
 * rows of the shard whose canonical key (score desc, id asc) beats the k-th entry of the query's answer in
 * d_ids / d_scores ([256][k] rows of the batch).  d_counts[i] includes the k - 1 better entries of the answer itself
 * when they are real rows, so an exact answer gives d_counts[i] == (number of valid entries among its first k - 1).
 * A row is read once for all nq queries: the whole batch of 256 is checked against an exact scan in 32 passes.
 * d_counts has TWENTY-FOUR words (ABI 400; it was sixteen): d_counts[16 + i] = rows at or above the k-th entry that were
 * not looked up because one thread had met 64 of them for that query already (> 0 means the pair count is a lower bound:
 * "check truncated", not "answer wrong").  d_counts[8 + i] = the rows at or above the k-th entry whose (id, canonical score) pair is an
 * entry of the answer BIT FOR BIT — an exact answer gives d_counts[8 + i] == (number of valid entries among all k): an entry
 * with a wrong score, a wrong id or a row that does not reach the k-th key is not counted.  (An all-padding answer, k-th id
 * -1, is not pair-checked: d_counts[8 + i] stays 0.)
 * row_format: 0 fp16 rows, 1 fp8 rows + d_row_scale, 2 fp32 rows.
 */
int rarc_verify_batch(const void* d_rows, const float* d_row_scale, int row_format, int64_t n_rows, int d_pad,
                      const void* d_qblock, int q_first, int nq, int k, int64_t id_base, const int64_t* d_ids,
                      const float* d_scores, uint32_t* d_counts, void* stream);

/*
 * Optional int8 "shadow" image of an fp16 corpus.  The int8-prefilter scan converts every fp16 row to
 * int8 on each search; with d_pad a multiple of 256 the conversion can be done once, at ingest
 * (rarc_quant_shadow_f16 = rarc_quant_meta_f16 + the image, int8 [ceil32(n_rows)][d_pad], caller-owned),
 * and rarc_search_f16_shadow then scans the image instead: half the HBM traffic per search and no
 * conversion arithmetic, for 50 % more HBM footprint.  Candidates are identical by construction (same
 * bytes, same bound), the canonical rescore still reads the fp16 rows: results are bit-identical to
 * rarc_search_f16.
 */
int rarc_quant_shadow_f16(const uint16_t* d_corpus_f16, int64_t n_rows, int d_pad, int64_t first_row,
                          float* d_qmeta, int8_t* d_shadow8, void* stream);
int rarc_search_f16_shadow(const uint16_t* d_corpus_f16, const int8_t* d_shadow8, int64_t n_rows, int d_pad,
                           const float* d_qmeta, const void* d_qblock, int nq, int k, int kprime,
                           int64_t id_base, float bin_lo, float bin_hi, int64_t* d_out_ids,
                           float* d_out_scores, uint32_t* d_status, void* d_workspace, size_t workspace_bytes,
                           int cand_cap, void* stream);

/*
 * fp8 corpus (BASELINE.json config 5's storage): rows of OCP e4m3fn bytes, d_pad a multiple of 256
 * (rarc_padded_dim_f8), plus one fp32 scale per row: value[m] = d_row_scale[r] * decode(byte[m]).
 * Half the HBM footprint and scan traffic of fp16.  The calls mirror their fp16 counterparts:
 *   rarc_ingest_f8      normalise (as rarc_ingest_f16), scale = max|x| / 448, byte = encode(x / scale)
 *                       (round to nearest even, saturating); d_row_norm2 optional
 *   rarc_quant_meta_f8  metadata for the int8-prefilter scan: rarc_quant_meta_floats_f8(n_rows) floats,
 *                       [0] = R as for fp16, then per 32-row tile: scale, 1/scale, 32 row multipliers
 *   rarc_search_f8      int8-prefilter scan + canonical finalize; the canonical score of a row is
 *                       d_row_scale[r] * (canonical fp32 dot of the query with the decoded bytes)
 *   rarc_repair_f8      exact single-query repair / verification
 * The query block is the one rarc_prep_queries writes (same d_pad; pass the fp8 d_qmeta).
 */
int rarc_padded_dim_f8(int d);
int rarc_ingest_f8(const float* d_in, int64_t ld_in, uint8_t* d_corpus_f8, int d_pad, float* d_row_scale,
                   float* d_row_norm2, int64_t n_rows, int d, int normalize, void* stream);
size_t rarc_quant_meta_floats_f8(int64_t n_rows);
int rarc_quant_meta_f8(const uint8_t* d_corpus_f8, const float* d_row_scale, int64_t n_rows, int d_pad,
                       int64_t first_row, float* d_qmeta, void* stream);
int rarc_search_f8(const uint8_t* d_corpus_f8, const float* d_row_scale, int64_t n_rows, int d_pad,
                   const float* d_qmeta, const void* d_qblock, int nq, int k, int kprime, int64_t id_base,
                   float bin_lo, float bin_hi, int64_t* d_out_ids, float* d_out_scores, uint32_t* d_status,
                   void* d_workspace, size_t workspace_bytes, int cand_cap, void* stream);
int rarc_repair_f8(const uint8_t* d_corpus_f8, const float* d_row_scale, int64_t n_rows, int d_pad,
                   const void* d_qblock, int q, int k, int64_t id_base, int64_t* d_out_ids, float* d_out_scores,
                   uint32_t* d_found, void* d_workspace, size_t workspace_bytes, void* stream);

/*
 * fp32 corpus — the reference's own storage (np.float32 rows in IndexFlatIP, VectorStore_Faiss.py:170,199-202):
 * returned scores are canonical fp32 dots of the fp32 query with the fp32 rows, no storage rounding at all
 * (what SURVEY.md §8(b) lists as the fp32 variant of rarc_score_topk).  The scan kernels still stream a 2-byte
 * image: the index holds fp32 rows [ceil32(n)][d_pad] for the rescore AND their fp16 image for the scan (6 bytes per
 * element); candidates are discarded on the image's int8 scores under a bound that includes the image's rounding.
 *   rarc_ingest_f32  normalise exactly as rarc_ingest_f16, store the fp32 row (zero padded) and its fp16 image
 *   metadata         rarc_quant_meta_f16 on the IMAGE; then the caller writes d_qmeta[1] = rho >= max ||d32 - d16||
 *                    (2^-11 * max||d|| + 2^-25 * sqrt(d_pad) covers normal and subnormal halves)
 *   rarc_search_f32  int8-prefilter scan of the image + canonical finalize on the fp32 rows
 *   rarc_repair_f32  exact single-query repair / verification on the fp32 rows
 */
int rarc_ingest_f32(const float* d_in, int64_t ld_in, float* d_corpus_f32, uint16_t* d_image_f16, int d_pad,
                    float* d_row_norm2, int64_t n_rows, int d, int normalize, void* stream);
int rarc_search_f32(const float* d_corpus_f32, const uint16_t* d_image_f16, int64_t n_rows, int d_pad,
                    const float* d_qmeta, const void* d_qblock, int nq, int k, int kprime, int64_t id_base,
                    float bin_lo, float bin_hi, int64_t* d_out_ids, float* d_out_scores, uint32_t* d_status,
                    void* d_workspace, size_t workspace_bytes, int cand_cap, void* stream);
/*
 * ABI 600 — one batch in one call: rarc_prep_queries + the rarc_search_* of the row format, i.e. the whole of
 *   VectorStore_Faiss.py:258-263 (`np.array([embedding]).astype(float32)`, `_normalize_vectors`, `index.search`)
 * for up to 256 queries, with what the two-call form needs a launch each for folded into the kernels that run anyway:
 *   d_status   RARC_MAX_QUERIES + 1 words; ZEROED by the query-prep kernel (the two-call form expects zeros from its caller)
 *   flag_host  pinned host word or NULL: zeroed by the query-prep kernel and set non-zero by the finalize kernel when any query
 *              is flagged — what a host otherwise copies out of d_status[RARC_MAX_QUERIES] behind the search; read it after an
 *              event recorded behind this call has completed
 *   gate_event hipEvent_t or NULL: the SCAN waits for it, the query prep and the seed pass in front of the scan do not.  Two
 *              search contexts (each its own query block, workspace and stream) that gate their scans on each other's
 *              completion run one batch's prep / seed under the other's finalize instead of behind it (config 2: 5 % per batch).
 * row_format 0: fp16 rows (d_aux NULL) · 1: fp8 rows, d_aux = float row scales · 2: fp32 rows, d_aux = their fp16 image ·
 *            3: fp16 rows, d_aux = int8 shadow image.  Everything else as in rarc_prep_queries / rarc_search_*.
 */
typedef struct RarcSearchBatch {
  const void* d_rows; const void* d_aux; int row_format; int64_t n_rows; int d_pad; const float* d_qmeta;
  const float* d_queries; int64_t ld_queries; int nq; int d; int normalize; float corpus_max_norm; void* d_qblock;
  int k, kprime; int64_t id_base; float bin_lo, bin_hi;
  int64_t* d_out_ids; float* d_out_scores; uint32_t* d_status; uint32_t* flag_host;
  void* d_workspace; size_t workspace_bytes; int cand_cap;
  void* gate_event;
} RarcSearchBatch;
int rarc_search_batch(const RarcSearchBatch* batch, void* stream);

/*
 * Measurement helper (ABI 600): stream n_bytes of device memory through every CU once (16-byte loads, eight in flight per lane,
 * one persistent workgroup per CU) and return nothing but a checksum — the READ ceiling of this part's HBM as a kernel
 * of the scan's shape can reach it, which bench.py reports next to the nominal 8 TB/s ("report against both").
 * d_sink: 8 bytes of device memory.  Time it with events on `stream`.
 */
int rarc_stream_read(const void* d_src, size_t n_bytes, void* d_sink, void* stream);

int rarc_repair_f32(const float* d_corpus_f32, int64_t n_rows, int d_pad, const void* d_qblock, int q, int k,
                    int64_t id_base, int64_t* d_out_ids, float* d_out_scores, uint32_t* d_found, void* d_workspace,
                    size_t workspace_bytes, void* stream);

/*
 * Merge G sorted candidate lists per query into the global top-k
 * (score desc, id asc).  Used after the RCCL all-gather of per-shard results
 * (SURVEY.md §8e).  Inputs are [G][nq][k]; outputs [nq][k].
 */
int rarc_topk_merge(const int64_t* d_ids, const float* d_scores, int n_lists, int nq, int k,
                    int64_t* d_out_ids, float* d_out_scores, void* stream);
/*
 * The same merge over lists in the form that crosses the all-gather as ONE tensor: int32 [n_lists][nq][k][3] =
 * (id low word, id high word, fp32 score bits).  rarc_pack_results writes one rank's [nq][k][3] block.
 */
int rarc_pack_results(const int64_t* d_ids, const float* d_scores, int nq, int k, int32_t* d_packed, void* stream);
int rarc_topk_merge_packed(const int32_t* d_packed, int n_lists, int nq, int k, int64_t* d_out_ids,
                           float* d_out_scores, void* stream);

/*
 * Reciprocal-rank fusion, bit-exact with RRFusion.fuse at core/utils/Fusion.py:45-76.
 * For query b, list r holds d_len[b*n_lists + r] keys at
 * d_keys[(b*n_lists + r)*max_len ...] in rank order (rank = position + 1).
 * score(key) = sum over occurrences, in (list, position) order, of
 * 1.0 / (rrf_k + rank) in fp64; output order = score desc, ties in first-insertion
 * order; d_out_n[b] = number of fused entries written (<= top_k).
 */
int rarc_rrf_fuse(const int64_t* d_keys, const int32_t* d_len, int nq, int n_lists, int max_len,
                  double rrf_k, int top_k, int64_t* d_out_keys, double* d_out_scores,
                  int32_t* d_out_n, void* stream);

/*
 * Reranker score -> order step of Qwen3Reranker (core/rerank/Reranker_Qwen3.py:41-49, :70-74):
 * p_yes = exp(log_softmax([z_no, z_yes])[1]) evaluated the way the reference's
 * fp16 tensors evaluate it, then a stable descending sort.  Inputs are fp16
 * logits [nq][n]; outputs: fp16 scores (as uint16 bit patterns) and the
 * permutation (int32) per query.
 */
int rarc_rerank_order(const uint16_t* d_z_no, const uint16_t* d_z_yes, int nq, int n,
                      uint16_t* d_out_scores_f16, int32_t* d_out_perm, void* stream);

/*
 * Maximal-marginal-relevance selection — the greedy loop of _mmr_select (VectorStore_Faiss.py:16-62) over candidates
 * that are already on the device (the resident rows of the fetch_k nearest neighbours, gathered: no re-embedding).
 * float64 arithmetic; candidate 0 first, then k - 1 rounds of  lambda·<q, e_i> − (1 − lambda)·max(0, max_sel <e_s, e_i>),
 * first maximum wins (python's max).  normalize != 0: query and candidates are divided by their norms first (cosine
 * stores, :308-312).  d_out: min(k, n) candidate indices in selection order.  d_work: rarc_mmr_workspace_doubles(n, d).
 */
size_t rarc_mmr_workspace_doubles(int n, int d);
int rarc_mmr_select(const float* d_cand, int64_t ld, const double* d_query, int n, int d, int normalize, int k,
                    double lambda, double* d_work, int32_t* d_out, void* stream);

/*
 * Deterministic synthetic corpus / query generator (bench + full-size property
 * tests): counter-based integer hash -> 52-bit uniform -> N(0,1) by the inverse normal CDF (Wichura AS 241,
 * exactly rounded double operations only) -> integer on a 2^-20 grid -> exact
 * integer sum of squares -> fp64 scale -> fp16 (RNE).  Bit-identical to
 * oracle/rarc_oracle.c:synth_rows_f16.  Row r of the stream `seed` depends only
 * on (seed, first_row + r, d).
 */
int rarc_synth_rows_f16(uint16_t* d_out_f16, int d_pad, int d, int64_t first_row, int64_t n_rows,
                        uint64_t seed, void* stream);
int rarc_synth_rows_f32(float* d_out_f32, int64_t ld_out, int d, int64_t first_row,
                        int64_t n_rows, uint64_t seed, void* stream);

/*
 * Encoder forward (BERT family) — what HuggingFaceEmbeddings reaches through
 * sentence-transformers at core/file_management/embeddings/huggingface.py:122-126
 * ([external] BERT forward -> CLS pooling -> optional normalise).  Token ids in, embeddings out.
 * All tensors fp16 (uint16 bit patterns) unless noted; weights use torch.nn.Linear layout [out][in].
 *   rarc_enc_embed_ln : out[t] = LayerNorm(word[ids[t]] + pos[t % seq_len] + type0); `vocab` = rows of d_word:
 *                       an id outside [0, vocab) is clamped into the table (memory safety only — callers
 *                       validate ids on the host, as the python binding does)
 *   rarc_enc_gemm     : C[M][N] = A[M][K] · W[N][K]^T + bias[N], act 0 = none, 1 = erf-GELU, 3 = SwiGLU over gate/up
 *                       columns interleaved in groups of 8 (C is then [M][N/2]; large shapes only, see RarcLmLayer);
 *                       M, N multiples of 128, K multiple of 64 (MFMA 32x32x16 f16, fp32 accumulate)
 *   rarc_enc_attention: ctx = softmax(Q K^T / sqrt(dh) + key mask) V per (sequence, head) from the
 *                       fused qkv [n_seq*seq_len][3*hidden]; keys >= d_lens[seq] are masked; dh 32 or 64
 *   rarc_enc_add_ln   : out = LayerNorm(x + resid)
 *   rarc_enc_pool     : out[b] = hidden[b*seq_len + 0] as fp32 (optionally L2-normalised)          [CLS pooling]
 *   rarc_enc_pool_mean: out[b] = mean over the d_lens[b] real tokens of hidden[b*seq_len + t], fp32  [mean pooling]
 */
int rarc_enc_embed_ln(const int32_t* d_ids, const uint16_t* d_word, const uint16_t* d_pos,
                      const uint16_t* d_type0, const uint16_t* d_gamma, const uint16_t* d_beta, float eps,
                      int n_tokens, int seq_len, int hidden, int vocab, uint16_t* d_out, void* stream);
int rarc_enc_gemm(const uint16_t* d_a, const uint16_t* d_w, const uint16_t* d_bias, uint16_t* d_c, int m,
                  int n, int k, int act, void* stream);
/* The same product for a caller whose bias is KNOWN to be zeros (the reranker LM's projections have none): d_zero_bias holds
 * n zeros; act 0 or 3.  Lets the large shapes run the tile kernel without a seam between output tiles (round 3), which has
 * no bias path; every other shape is rarc_enc_gemm. */
int rarc_enc_gemm_zero_bias(const uint16_t* d_a, const uint16_t* d_w, const uint16_t* d_zero_bias, uint16_t* d_c, int m,
                            int n, int k, int act, void* stream);
int rarc_enc_attention(const uint16_t* d_qkv, const int32_t* d_lens, int n_seq, int seq_len, int hidden,
                       int n_heads, uint16_t* d_ctx, void* stream);
int rarc_enc_add_ln(const uint16_t* d_x, const uint16_t* d_resid, const uint16_t* d_gamma,
                    const uint16_t* d_beta, float eps, int n_rows, int hidden, uint16_t* d_out, void* stream);
int rarc_enc_pool(const uint16_t* d_hidden, int n_seq, int seq_len, int hidden, int normalize,
                  float* d_out, void* stream);
int rarc_enc_pool_mean(const uint16_t* d_hidden, const int32_t* d_lens, int n_seq, int seq_len, int hidden,
                       int normalize, float* d_out, void* stream);

/*
 * Whole encoder forward in one call: the launch loop over the layers runs on the host side of this
 * library, not in the caller (at small batches the per-call overhead of a foreign-function binding is
 * longer than the kernels).  `model` and `model->layers` are HOST structs of DEVICE pointers.
 * `normalize`: bit 0 = L2-normalise the pooled vector, bit 1 = mean pooling over the real tokens instead of CLS.
 * n_seq*seq_len must be a multiple of 128 (pad seq_len to a multiple of 32 with masked tokens and n_seq to a
 * multiple of 4 with sequences of length 1); d_lens[s] in [1, seq_len] (clamped into that range).  d_ws: scratch of
 * rarc_enc_workspace_bytes(hidden, inter, n_seq*seq_len) bytes.  d_out: fp32 [n_seq][hidden].
 */
typedef struct RarcEncLayer {
  const uint16_t *qkv_w, *qkv_b; /* fused [3*hidden][hidden], [3*hidden] (query, key, value) */
  const uint16_t *o_w, *o_b, *ln1_g, *ln1_b;
  const uint16_t *f1_w, *f1_b, *f2_w, *f2_b, *ln2_g, *ln2_b;
} RarcEncLayer;
typedef struct RarcEncModel {
  int hidden, heads, inter, n_layers;
  float ln_eps;
  const uint16_t *word, *pos, *type0, *emb_g, *emb_b;
  const RarcEncLayer* layers; /* host array [n_layers] */
  int vocab;   /* rows of `word`  (REQUIRED > 0: token ids are clamped into [0, vocab)) */
  int max_pos; /* rows of `pos`   (REQUIRED >= seq_len) */
  /* MPNet family (the reference's default checkpoint, sentence-transformers/all-mpnet-base-v2, huggingface.py:6): one
   * relative-position bias shared by all layers, added to the scaled scores before the softmax
   * (transformers MPNetEncoder.compute_position_bias).  It depends on key - query only:
   * rel_bias fp32 [heads][2*rel_span - 1], entry [h][key - query + rel_span - 1]; null = none (BERT).
   * REQUIRED rel_span >= seq_len when set.  (MPNet's position ids start at 2 and it has no token types: pass
   * pos = table + 2 rows, max_pos = rows - 2, type0 = zeros.) */
  const float* rel_bias;
  int rel_span;
} RarcEncModel;
size_t rarc_enc_workspace_bytes(int hidden, int inter, int n_tokens);
int rarc_enc_forward(const RarcEncModel* model, const int32_t* d_ids, const int32_t* d_lens, int n_seq,
                     int seq_len, int normalize, void* d_ws, size_t ws_bytes, float* d_out, void* stream);

/*
 * The same forward at the REFERENCE's precision (fp32-class; `precision = "fp32"` of the embedding provider).
 * `SentenceTransformer(model_name, **model_kwargs)` (huggingface.py:96-98) loads fp32 weights and `.encode` (:122-126)
 * runs an fp32 forward; rarc_enc_forward above is an fp16 approximation of it (1e-3 class), this one agrees with an
 * fp32 forward to fp32 rounding noise.  The GEMMs run on the fp16 MFMA over SPLIT operands: x*s = hi + lo (two fp16
 * numbers, s a power of two per row, together 22 bits of x), C = [A_lo|A_hi|A_hi] * [W_hi|W_lo|W_hi]^T / (s_a s_w): one
 * fp16 GEMM over K' = 3K with fp32 accumulation.  Residual stream, LayerNorm, GELU (libm erff), softmax (libm expf),
 * attention products and pooling are fp32.
 *   rarc_enc32_split_weight : W fp32 [n][k] (torch Linear layout) -> d_w3 fp16 [n][3k] = [hi | lo | hi], d_rw fp32 [n] = 1/s_w
 *   rarc_enc32_split_rows   : X fp32 [m][k] -> d_a3 fp16 [m][3k] = [lo | hi | hi], d_ra fp32 [m] = 1/s_a   (k % 4 == 0, k <= 4096)
 *   rarc_enc32_gemm         : C fp32 [m][n] = X W^T + bias from the two split images (m, n multiples of 128, k of 64)
 *   rarc_enc32_forward      : token ids -> fp32 embeddings; arguments as rarc_enc_forward; hidden <= 1024, inter <= 4096
 */
typedef struct RarcEnc32Layer {
  const uint16_t* qkv_w3; const float *qkv_rw, *qkv_b;           /* fused [3*hidden][3*hidden] split rows, [3*hidden], [3*hidden] */
  const uint16_t* o_w3;   const float *o_rw, *o_b, *ln1_g, *ln1_b;
  const uint16_t* f1_w3;  const float *f1_rw, *f1_b;
  const uint16_t* f2_w3;  const float *f2_rw, *f2_b, *ln2_g, *ln2_b;
  const float* f1_colmax;   /* ABI 401.  float [hidden + 1]: c[k] = max_j |W1[j][k]| of the fp32 FFN1 weight, c[hidden] = max_j |b1[j]|;
                               NULL = none.  With it, batches that fill the chip run FFN1 with bias + GELU + the split of its output
                               fused into the GEMM (the row's power-of-two scale comes from the bound sum_k |x_k| c[k] + c[hidden]
                               >= max_j |gelu(x·W1_j + b1_j)|, known before the GEMM); without it FFN1 is a product + a row pass. */
  /* ABI 600 — the QUERY PATH's weight images (rarc_enc32_pack_query_weight), one per projection, NULL = none.  With all four
   * present in every layer, a forward of 32 / 64 / 128 tokens (one to four 32-token queries: the reference's embed_query,
   * core/file_management/embeddings/huggingface.py:136-145) runs its projections as weight streams over these images instead
   * of the 128 x 128 tile kernels over *_w3.  uint16 (fp16 bits) [n / 32][k / 16][2][64][8]: per 32 output features and 16-wide
   * k step the W_hi fragment, then the W_lo fragment, each in the lane order of a v_mfma_f32_32x32x16_f16 A operand. */
  const uint16_t *qkv_wq, *o_wq, *f1_wq, *f2_wq;
} RarcEnc32Layer;
typedef struct RarcEnc32Model {
  int hidden, heads, inter, n_layers;
  float ln_eps;
  const float *word, *pos, *type0, *emb_g, *emb_b; /* fp32 tables and LayerNorm parameters */
  const RarcEnc32Layer* layers;                    /* host array [n_layers] */
  int vocab, max_pos;                              /* REQUIRED, as in RarcEncModel */
  const float* rel_bias;                           /* relative-position attention bias, as in RarcEncModel (null = none) */
  int rel_span;
} RarcEnc32Model;
int rarc_enc32_split_weight(const float* d_w, int n, int k, uint16_t* d_w3, float* d_rw, void* stream);
/* d_w3: the split rows of an [n][k] weight (rarc_enc32_split_weight) -> d_wq: n*k*2 halves, the query path's image of the same
 * weight (RarcEnc32Layer.*_wq).  n a multiple of 32, k a multiple of 128. */
int rarc_enc32_pack_query_weight(const uint16_t* d_w3, int n, int k, uint16_t* d_wq, void* stream);
int rarc_enc32_split_rows(const float* d_x, int m, int k, uint16_t* d_a3, float* d_ra, void* stream);
int rarc_enc32_gemm(const uint16_t* d_a3, const float* d_ra, const uint16_t* d_w3, const float* d_rw,
                    const float* d_bias, float* d_c, int m, int n, int k, void* stream);
size_t rarc_enc32_workspace_bytes(int hidden, int inter, int n_tokens);
int rarc_enc32_forward(const RarcEnc32Model* model, const int32_t* d_ids, const int32_t* d_lens, int n_seq,
                       int seq_len, int normalize, void* d_ws, size_t ws_bytes, float* d_out, void* stream);

/*
 * Reranker LM forward — what Qwen3Reranker.compute_logits takes from the causal LM
 * (core/rerank/Reranker_Qwen3.py:41-49: `self.lm(**inputs).logits[:, -1, :]` at the ids of "no" and "yes") for
 * the LEFT-padded batches process_inputs builds (:29-39).  Qwen3-style decoder: RMSNorm, fused q|k|v projection
 * with grouped K/V heads, per-head q/k RMSNorm, rotary embedding (rotate-half), causal attention, SwiGLU MLP, no
 * biases; fp16 weights (torch.nn.Linear layout [out][in]) and activations, fp32 accumulation.
 *   d_ids    int32 [n_seq][seq_len], left padded;  d_start int32 [n_seq]: index of each sequence's first real token
 *            (positions run 0..seq_len-1 over the padded sequence, as in the reference's forward)
 *   d_out    fp16 [n_seq][2]: (logit of no_id, logit of yes_id) at the last position — the input of rarc_rerank_order
 * n_seq*seq_len a multiple of 128; head_dim 64 or 128; hidden, inter, (n_q+2*n_kv)*head_dim multiples of 128.
 * `model`, `model->layers` are HOST structs of DEVICE pointers; zero_bias: max(hidden, 2*inter, qkv width) zeros.
 */
typedef struct RarcLmLayer {
  const uint16_t *in_norm, *qkv_w, *q_norm, *k_norm, *o_w, *post_norm, *gate_up_w, *down_w;
  /* OPTIONAL (null = not supplied; ABI 320): qkv_w with its columns multiplied by in_norm, gate_up_w with its columns
   * multiplied by post_norm (products in fp32, rounded to fp16 once).  With both present, batches large enough for the 256-row
   * tile kernels run WITHOUT the RMSNorm passes: RMSNorm(x)·Wᵀ = r ⊙ (x·(W ⊙ γ)ᵀ) with r = rsqrt(mean(x²) + eps) per row, so
   * the projection reads the residual stream itself and scales its output rows, and the output / down projections add into
   * the residual stream in their epilogue.  Same function, one rounding moved: the reference rounds the normed
   * activations to fp16 before the product. */
  const uint16_t *qkv_w_folded, *gate_up_w_folded;
  /* qkv_w [(n_q+2*n_kv)*head_dim][hidden] = q_proj | k_proj | v_proj rows;
   * gate_up_w [2*inter][hidden] = gate_proj and up_proj rows INTERLEAVED in groups of 8: rows 16b .. 16b+7 are
   * gate_proj rows 8b .. 8b+7, rows 16b+8 .. 16b+15 the up_proj rows of the same features (so that one lane of the
   * GEMM's 32 x 32 MFMA blocks holds gate and up of a feature and the SwiGLU is its epilogue; inter % 8 == 0) */
} RarcLmLayer;
typedef struct RarcLmModel {
  int hidden, n_layers, n_q_heads, n_kv_heads, head_dim, inter, vocab;
  float rms_eps, rope_theta;
  const uint16_t *embed, *lm_head, *final_norm, *zero_bias;
  const RarcLmLayer* layers; /* host array [n_layers] */
} RarcLmModel;
size_t rarc_lm_workspace_bytes(const RarcLmModel* model, int n_tokens);
int rarc_lm_yes_no_logits(const RarcLmModel* model, const int32_t* d_ids, const int32_t* d_start, int n_seq,
                          int seq_len, int no_id, int yes_id, void* d_ws, size_t ws_bytes, uint16_t* d_out_f16,
                          void* stream);

/*
 * Shared prompt prefixes.  The prompts Qwen3Reranker.rerank scores for ONE query (Reranker_Qwen3.py:23-27,57-66) are
 * identical up to the document text — chat prefix, instruction, query: ~80 of ~220 tokens — and in a causal LM the keys and
 * values of those tokens do not depend on what follows them.  rarc_lm_prefix_kv runs the n_prefix LEFT-padded prefixes
 * [n_prefix][prefix_len] once and stores every layer's raw k | v rows in d_cache (rarc_lm_prefix_cache_bytes(model,
 * n_prefix * prefix_len) bytes); rarc_lm_yes_no_logits_prefixed then runs only the REMAINDERS d_ids [n_seq][seq_len] (left
 * padded), sequence s attending to the cache rows of prefix d_prefix_of[s] (from d_prefix_start[prefix]) and then to its own
 * tokens; rotary positions continue from prefix_len.  Same logits as rarc_lm_yes_no_logits on the concatenated prompts up to
 * fp16 rounding (rotary embeddings are relative; the softmax sums run in a different order).  n_prefix * prefix_len and
 * n_seq * seq_len multiples of 128.  d_prefix_of[s] = -1: no prefix for s.
 */
size_t rarc_lm_prefix_cache_bytes(const RarcLmModel* model, int n_prefix_tokens);
int rarc_lm_prefix_kv(const RarcLmModel* model, const int32_t* d_ids, const int32_t* d_start, int n_prefix, int prefix_len,
                      void* d_ws, size_t ws_bytes, void* d_cache, size_t cache_bytes, void* stream);
int rarc_lm_yes_no_logits_prefixed(const RarcLmModel* model, const int32_t* d_ids, const int32_t* d_start, int n_seq,
                                   int seq_len, const int32_t* d_prefix_of, const void* d_cache, int n_prefix,
                                   int prefix_len, const int32_t* d_prefix_start, int no_id, int yes_id, void* d_ws,
                                   size_t ws_bytes, uint16_t* d_out_f16, void* stream);

/*
 * Batch WordPiece tokenisation on the host (no device involved) — what sits between the texts the reference hands to
 * `SentenceTransformer.encode` (core/file_management/embeddings/huggingface.py:116-126) and the encoder forward above: the
 * BERT tokeniser (native and multi-threaded in the reference's dependency too).  Same algorithm as the python restatement
 * encapsulation/embeddings/wordpiece.py (pinned to transformers.BertTokenizer), for ASCII texts, on n_threads cores.
 *   rarc_wordpiece_create   vocab_blob: the vocabulary's tokens separated by '\n', id = position (a vocab.txt image);
 *                           never_split_blob: the SPECIAL tokens, '\n'-separated — cut out of the raw text wherever they
 *                           occur, before anything else (unk / cls / sep / pad / mask and whatever else the caller lists);
 *                           the handle is the one object this library allocates — release it with _destroy
 *   rarc_wordpiece_encode   text i = bytes [text_offsets[i], text_offsets[i+1]) of text_blob;  h_ids [n_texts][ld_ids] int32
 *                           receives [CLS] ids[: max_length - 2] [SEP] padded to max_length with the pad id, h_lens[i] the
 *                           number of real ids — or -1 for a text with a byte >= 0x80, which is NOT tokenised here (its
 *                           Unicode normalisation steps are left to the python tokeniser; row i is then untouched)
 */
typedef struct RarcWordPiece RarcWordPiece;
int rarc_wordpiece_create(const char* vocab_blob, size_t blob_bytes, int do_lower_case, const char* unk_token,
                          const char* cls_token, const char* sep_token, const char* pad_token, const char* never_split_blob,
                          size_t never_split_bytes, int max_input_chars_per_word, RarcWordPiece** out);
void rarc_wordpiece_destroy(RarcWordPiece* wp);
int rarc_wordpiece_encode(const RarcWordPiece* wp, const char* text_blob, const int64_t* text_offsets, int n_texts,
                          int max_length, int32_t* h_ids, int64_t ld_ids, int32_t* h_lens, int n_threads);

/*
 * Shard files: bulk movement of stored rows between a file and HBM — the native I/O behind save_local / load_local,
 * counterpart of faiss.write_index / faiss.read_index at
 *   encapsulation/database/vector_db/VectorStore_Faiss.py:438 (save_local :432-450) and :467 (load_local :452-482)
 * for an index whose rows are resident in HBM.  A shard never exists as a host array: both calls stream
 * `n_seg` byte ranges (file offset h_file_off[s], h_bytes[s] bytes <-> device offset h_dev_off[s] from d_base) through a
 * ring of pinned host slots cut from h_staging (caller-owned PINNED memory, 4096-byte aligned; 2 * n_threads slots of
 * staging_bytes / (2 n_threads) bytes, >= 64 KiB each).  n_threads workers each alternate file transfer and DMA (two slots
 * per worker; the DMA is enqueued on `stream`), so disk and PCIe overlap and the host footprint is the ring whatever the
 * shard size.
 *   RARC_IO_DIRECT  also open the file O_DIRECT: chunks whose offset, length and slot are 4096-aligned bypass the page
 *                   cache (storage DMA -> pinned slot -> GPU DMA); a file system without O_DIRECT is served buffered
 *   RARC_IO_FSYNC   rarc_device_to_file: fsync before returning
 *   RARC_IO_TRUNCATE rarc_device_to_file: cut the file at the end of the furthest segment first (a caller overwriting a
 *                   LONGER file in one call; without it the file is never truncated: the .rarc writer sizes the file, writes
 *                   its header and small sections itself and lets this call fill the row section — DESIGN.md 3)
 * rarc_device_to_file creates the file if needed; rarc_file_to_device fails if the file is shorter than a segment.  File
 * errors (open, read / write, fsync: ENOSPC, EIO ...) return RARC_E_IO, HIP errors RARC_E_HIP.
 * Both are synchronous (the bytes are in place on return; `stream` is drained).  d_capacity_bytes bounds every
 * segment's device range.  `stats` (optional) receives what was moved and how fast.
 */
#define RARC_IO_DIRECT 1
#define RARC_IO_FSYNC 2
#define RARC_IO_TRUNCATE 4
typedef struct RarcIoStats {
  int64_t bytes;            /* bytes moved */
  double seconds;           /* wall time of the call's transfer phase */
  double file_seconds;      /* pread / pwrite time summed over the workers */
  double copy_wait_seconds; /* time the workers waited for their DMA, summed */
  int64_t direct_bytes;     /* of `bytes`, how many went through O_DIRECT */
  int64_t n_chunks, slot_bytes;
  int n_threads, direct;    /* direct: 1 if the O_DIRECT descriptor could be opened */
} RarcIoStats;
int rarc_file_to_device(const char* path, int n_seg, const int64_t* h_file_off, const int64_t* h_bytes,
                        const int64_t* h_dev_off, void* d_base, int64_t d_capacity_bytes, void* h_staging,
                        size_t staging_bytes, int n_threads, int flags, void* stream, RarcIoStats* stats);
int rarc_device_to_file(const char* path, int n_seg, const int64_t* h_file_off, const int64_t* h_bytes,
                        const int64_t* h_dev_off, const void* d_base, int64_t d_capacity_bytes, void* h_staging,
                        size_t staging_bytes, int n_threads, int flags, void* stream, RarcIoStats* stats);

/*
 * Exact top-k beyond what the register-resident scans take: rows wider than 1024 padded dimensions (up to 4096) and k up
 * to 8192 — faiss.IndexFlatIP takes any d and any k (encapsulation/database/vector_db/VectorStore_Faiss.py:101-115, :262-263;
 * the reference's other embedding source returns 1536- / 3072-d vectors, encapsulation/llm/openai_llm.py:139-161).
 * The score matrix goes through the encoder's MFMA GEMM one chunk of rows at a time (fp16 scores, never more than 64 MB of
 * them), a select pass nominates rows against a rising, rigorous threshold, and the finalize rescores the nominees with the
 * canonical fp32 inner product and orders them (score desc, id asc): ids and scores are the oracle's, bit for bit
 * (csrc/wide.hip has the bound).  fmt 0: fp16 rows; fmt 2: fp32 rows + their fp16 image (what the GEMM reads; rho >=
 * ||row - image||, as qmeta[1]).  d_qblock as written by rarc_prep_queries (which takes d_pad up to 4096).  d_status:
 * uint32 [256], a query whose candidate list filled up carries RARC_Q_OVERFLOW — call again with a larger cand_cap
 * (cand_cap >= n_rows cannot overflow; it must be at least max(16384, 2k) rounded up to 256, plus 256: the first chunk of rows — and a multiple of 8).
 */
size_t rarc_wide_workspace_bytes(int d_pad, int cand_cap);
int rarc_search_wide(const void* d_rows, const uint16_t* d_image16, int fmt, int64_t n_rows, int d_pad, float max_norm,
                     float rho, const void* d_qblock, int nq, int k, int64_t id_base, int64_t* d_out_ids,
                     float* d_out_scores, uint32_t* d_status, void* d_ws, size_t ws_bytes, int cand_cap, void* stream);

/*
 * All pairs (i < j) of n embeddings whose cosine reaches a threshold — the entity de-duplication of the reference's graph
 * store (encapsulation/database/graph_db/Base_Neo4j.py:538-583: sklearn.metrics.pairwise.cosine_similarity over every entity
 * embedding, then a python loop over i < j keeping similarity >= 0.95); SURVEY 8(f) rank 4.  The n x n matrix is never
 * formed: the rows are normalised into an fp16 image, the score GEMM of rarc_search_wide (select in its epilogue) nominates
 * the pairs whose approximate cosine reaches threshold - eps (eps bounds the fp16 / fp32-accumulation error), and each
 * nominated pair is scored exactly (double-precision dot product of the fp32 rows, double-precision norms) and kept if that
 * reaches the threshold (csrc/pairs.hip).
 * d_rows: fp32 [n_rows][ld], device, any scale; 1 <= d <= 4096; 0 < threshold <= 1.
 * Output, in no particular order: d_out_pairs int64 [out_cap][2] = (i, j), d_out_scores double [out_cap];
 * *d_out_count (uint64) = how many pairs reached the threshold, also when that exceeds out_cap.
 * *d_flags (uint32): bit 0 = a column's nomination list (cand_cap entries) overflowed — call again with a larger cand_cap
 * (cand_cap >= n_rows cannot overflow); bit 1 = more pairs than out_cap — call again with out_cap >= *d_out_count.
 */
size_t rarc_similar_pairs_workspace_bytes(int64_t n_rows, int d, int cand_cap);
int rarc_similar_pairs(const float* d_rows, int64_t ld, int64_t n_rows, int d, double threshold, void* d_ws, size_t ws_bytes,
                       int cand_cap, int64_t* d_out_pairs, double* d_out_scores, int64_t out_cap,
                       unsigned long long* d_out_count, uint32_t* d_flags, void* stream);

/*
 * Deleting rows of a resident index: stable in-place compaction.  The reference deletes by clearing the index and
 * embedding every surviving text again (encapsulation/database/vector_db/VectorStore_Faiss.py:374-415); the surviving
 * rows are in HBM already, so here they move down over the holes: row i of the result is the i-th surviving row.
 * d_rows: n_rows rows of row_bytes (a multiple of 4) — the fp16 / fp8 / fp32 rows, and likewise the fp8 row scales
 * (row_bytes 4), the fp32 index's fp16 image, the int8 shadow.  d_adj: device int64 [n_holes], d_adj[j] = h_j - j for
 * the sorted distinct hole rows h_j; first_hole = h_0.  d_tmp: caller-owned scratch (>= one row; the rows move through
 * it chunk by chunk).  The vacated tail [n_rows - n_holes, n_rows) is zeroed.  Tile metadata is NOT updated: call
 * rarc_quant_meta_* from first_hole afterwards.
 */
int rarc_compact_rows(void* d_rows, int64_t row_bytes, int64_t n_rows, const int64_t* d_adj, int64_t n_holes,
                      int64_t first_hole, void* d_tmp, size_t tmp_bytes, void* stream);

/*
 * Growable device arenas: where `index.add` appends to (encapsulation/database/vector_db/VectorStore_Faiss.py:199-202 —
 * faiss grows a std::vector there).  An arena owns VIRTUAL addresses for the largest size it may reach (no memory) and is
 * backed slab by slab as rows arrive: the base pointer never moves and nothing is copied, so the peak footprint of a
 * growing index is its live rows rounded up to one slab (a reallocating buffer holds old + new: up to 3x).
 * The second kind of object this library allocates (with the tokenizer handle): release it with rarc_vmem_destroy, after
 * the last kernel that reads it.  Arenas are slab-aligned sub-ranges of ONE address space the first create reserves for
 * the process and never frees (RARC_VMEM_SPACE_TIB TiB, default 16; addresses that were backed once are not handed out
 * again, so the space lasts for that many TiB of slabs mapped over the life of the process); every piece of physical memory is one SLAB
 * (slab_bytes, 0 = RARC_VMEM_DEFAULT_SLAB, rounded up to the device's mapping granularity), ONE slab size per process — a
 * create with another size returns RARC_E_UNSUPPORTED, a create the space has no room for RARC_E_WORKSPACE.
 * min_reserve_bytes (0 = reserve_bytes): when no free range holds reserve_bytes, the largest one that holds at least this
 * much is taken whole — rarc_vmem_reserved reports what the arena got.  (What this HIP
 * runtime does with anything else is written down in csrc/vmem.hip and reproducible with tools/vmem_probe.py.)
 * rarc_vmem_grow(min_bytes): back at least the first min_bytes, in whole slabs (never shrinks; on failure — HBM
 * exhausted — what was mapped stays mapped and usable).  Not tied to a stream: mapping is a host-side operation,
 * visible to later launches.
 */
#define RARC_VMEM_DEFAULT_SLAB (16u << 20)
typedef struct RarcVmem RarcVmem;
int rarc_vmem_create(int device, size_t reserve_bytes, size_t min_reserve_bytes, size_t slab_bytes, RarcVmem** out);
int rarc_vmem_grow(RarcVmem* arena, size_t min_bytes);
void* rarc_vmem_base(const RarcVmem* arena);
size_t rarc_vmem_mapped(const RarcVmem* arena);
size_t rarc_vmem_reserved(const RarcVmem* arena);
size_t rarc_vmem_slab(const RarcVmem* arena);
size_t rarc_vmem_granularity(const RarcVmem* arena);
int rarc_vmem_destroy(RarcVmem* arena);

/*
 * Measurement hooks (bench.py): while profiling is on, every rarc_search_f16 brackets its scan
 * kernel with a pair of HIP events recorded on the search's own stream.  rarc_profile_end
 * synchronises, returns the summed scan time and the number of launches measured, and releases
 * the events.  One profiling session at a time per process (the event list is guarded by a mutex, so searches
 * issued from other threads while a session is open are measured too, in launch order).
 */
int rarc_profile_begin(int max_launches);
int rarc_profile_end(double* total_scan_ms, int* n_launches);

#ifdef __cplusplus
}
#endif
#endif /* RARC_H */
