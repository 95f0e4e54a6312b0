/* ragroute_hip.h — C ABI of libragroute_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for RAGRoute's retrieval hot path.  The reference (sacs-epfl/ragroute) has
 * no FFI of its own: its hot path calls `faiss` / `torch.nn` / `numpy` from Python.  Each entry
 * point below replaces one of those numeric call sites; the reference file:line it stands in for
 * is cited per function.  All pointers named `d_*` / marked "device" are HIP device pointers;
 * `stream` is a `hipStream_t` passed as `void*` (NULL = default stream).  No function synchronises
 * the host with the device; results are valid once the stream has drained.  No function aborts:
 * every one returns RR_OK (0) or a negative rr_status and records a message for rr_last_error().
 */
#ifndef RAGROUTE_HIP_H
#define RAGROUTE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rr_status {
  RR_OK = 0,
  RR_ERR_INVALID = -1,     /* bad argument (shape, k, dtype, null pointer) */
  RR_ERR_UNSUPPORTED = -2, /* valid request this build has no kernel for */
  RR_ERR_WORKSPACE = -3,   /* workspace too small */
  RR_ERR_HIP = -4          /* a HIP runtime call or kernel launch failed */
} rr_status;

enum { RR_DTYPE_F16 = 0, RR_DTYPE_BF16 = 1 };
enum { RR_MAX_K = 1024, RR_QUERY_BLOCK = 256 };

/* Library version (major*10000 + minor*100 + patch). */
int rr_version(void);
/* The compiler flags this library was built with (ragroute_amd/_build.py).  A product build carries no -D switch: the
 * development kernels (RR_DEV_VARIANTS) and the timing-only ablations that exist inside such builds are then not compiled
 * in, and a test can tell the library under test from an A/B build.  A product build also reads NO environment variable:
 * the RR_* tuning variables (kernel selection, chunk schedule, sample size, timelines) are honoured by -DRR_DEV_VARIANTS
 * builds only. */
const char* rr_build_flags(void);
/* Message of the last failure on the calling thread ("" if none). */
const char* rr_last_error(void);
/* Number of compute units of the current device (the persistent scan grid), or <0 on error. */
int rr_device_cus(void);

/* Row width (in elements) the scan kernel needs for embedding dimension d: the corpus and the
 * queries must be stored with this leading dimension, zero padded: 128..768 in steps of 128 (queries stay
 * register-resident), 896 / 1024 / 1280 / 1536, above that the next multiple of 128 up to 8192 (wide-row kernel).  <0 if d is unsupported. */
int rr_padded_dim(int d);

/* In-place row-wise L2 normalisation of an f32 matrix, zero-norm rows left unchanged.
 * Replaces `faiss.normalize_L2(query_vec)` — reference ragroute/data_source.py:198-199
 * (also mmlu.py:105-106).  x: device f32 [n][d] contiguous. */
int rr_l2_normalize_f32(float* d_x, int64_t n, int64_t d, void* stream);

/* Ingest: convert f32 rows to the HBM-resident scan format (f16 or bf16, leading dimension
 * d_out >= d, zero padded), optionally L2-normalising each row first (cosine corpora).
 * Stands in for what `faiss.read_index` leaves in memory — reference
 * ragroute/data_source.py:69-80 — and is also used to prepare query batches
 * (data_source.py:113-114 builds the f32 [1,d] query).
 * d_x: device f32, n rows of d values, row stride ld_in; d_out: device [n][d_out] halves. */
int rr_rows_to_half(const float* d_x, int64_t n, int64_t d, int64_t ld_in, void* d_out, int dtype,
                    int64_t d_out_dim, int normalize, void* stream);

/* Column means of an HBM-resident corpus: the per-source "centroid" feature of the router.  The reference only
 * CONSUMES it (`corpus_stats["centroid"]`, ragroute/router.py:147-151; the *_stats.json files are produced
 * off-tree); this builds it for synthetic or new corpora.  d_xb as for rr_flat_search; d_out device f32 [dim]
 * (entries >= d are zero, matching the zero padding of router.py:150). */
int rr_centroid(const void* d_xb, int dtype, int64_t n_rows, int dim, int d, float* d_out, void* stream);

/* Bytes of device workspace rr_flat_search (and _l2, _segments) needs for this k on the current device: for rows of any
 * width, or — smaller for dim <= 768 — for rows of one padded width. */
size_t rr_flat_search_workspace_bytes(int k);
size_t rr_flat_search_workspace_bytes_for(int k, int dim);

/* Exact brute-force inner-product top-k of nq queries against an HBM-resident corpus.
 * Replaces `index.search(query_embed, k)` — reference ragroute/data_source.py:158, 186, 203
 * (med_rag.py:158, mmlu.py:109): returns, per query, the k best rows best-first; ties are
 * broken by ascending row id — THIS BUILD'S DEFINITION: which of several equally scored rows FAISS 1.7.4 lists first is
 * unverified (no faiss in the image; tests/test_faiss_parity_gpu.py and bench.py's faiss probe report
 * `ties_set_identical` / `ties_order_identical` wherever faiss is importable); if fewer than k rows exist the tail is
 * (-inf, -1) as FAISS pads.
 *   d_xb     device [n_rows][dim] f16/bf16, dim == rr_padded_dim(d) (zero padded)
 *   d_xq     device [nq][dim] same dtype
 *   d_D      device f32 [nq][k]  scores (inner products, f32 accumulation)
 *   d_I      device i64 [nq][k]  row ids + id_offset (shard base), or -1
 *   d_ws     device workspace of at least rr_flat_search_workspace_bytes(k) bytes
 *   d_route_mask  optional device u8, one byte per query at d_route_mask[q * mask_stride]: 0 means the router
 *            did not select this source for query q (router.py:282 keeps only selected sources; the front-end
 *            then never asks this data source, http_server.py:181-209) and the query's result is all padding.
 *            NULL = every query is served.
 * Any nq >= 0 is accepted (served in blocks of RR_QUERY_BLOCK); each query's result is the same
 * as that of a single-query call, as the reference issues them (data_source.py:114). */
int rr_flat_search(const void* d_xb, int dtype, int64_t n_rows, int dim, const void* d_xq, int nq, int k,
                   float* d_D, int64_t* d_I, int64_t id_offset, void* d_ws, size_t ws_bytes,
                   const uint8_t* d_route_mask, int64_t mask_stride, void* stream);

/* ONE search over several data sources that receive the same query embeddings (the four MedRAG sources all use MedCPT,
 * five FeB4RAG sources UAE-Large-V1: reference ragroute/config.py:37-71), instead of one `index.search` per source
 * (data_source.py:158, 186, 203) followed by the front-end's concatenation and `rerank_medrag` (http_server.py:280-293,
 * rerank.py:3-9): per query, the k best rows of the UNION of the segments the router selected for it, best-first — which is
 * what merging the per-source top-k lists yields.  One query preparation, one bootstrap and one chunk schedule serve all
 * sources; small sources no longer pay a search's fixed cost each.
 *   d_xb      ONE device matrix [n_rows_total][dim]; segment s = rows [row_begin, row_begin + n_rows) of it.  Segments are
 *             ascending, do not overlap and begin at multiples of RR_SEGMENT_ALIGN rows; rows between segments (alignment
 *             gaps) may hold anything finite or not — they are scanned but never returned.
 *   segs      HOST array of n_segs (<= RR_MAX_SEGMENTS) descriptors.  A segment is a whole source, or a ROW SLICE of one (a source
 *             cut over several GPUs, ragroute_amd/placement.py: several segments may then share a mask_col): its id_offset is
 *             (source << 40) + the slice's first row, so the ids it returns are the whole source's ids
 *   d_I       result ids: id_offset + (row - row_begin) of the row's segment, or -1.  Ties are broken by (segment order,
 *             row), i.e. by ascending id when the id_offsets ascend with the segments.
 *   d_route_mask  optional device u8 [nq][mask_stride]: segment s serves query q iff mask_col < 0 or
 *             d_route_mask[q * mask_stride + mask_col] != 0 (router.py:276-282); a query routed nowhere gets all padding.
 *   d_ws      as rr_flat_search (rr_flat_search_workspace_bytes(k)).  Inner product / cosine only. */
typedef struct rr_segment {
  int64_t row_begin;
  int64_t n_rows;
  int64_t id_offset;
  int32_t mask_col;
  int32_t reserved;
} rr_segment;
enum { RR_MAX_SEGMENTS = 32, RR_SEGMENT_ALIGN = 256 };
int rr_flat_search_segments(const void* d_xb, int dtype, int64_t n_rows_total, int dim, const rr_segment* segs, int n_segs,
                            const void* d_xq, int nq, int k, float* d_D, int64_t* d_I, void* d_ws, size_t ws_bytes,
                            const uint8_t* d_route_mask, int64_t mask_stride, void* stream);

/* The same search under the squared-L2 metric (the role of faiss.IndexFlatL2: the reference's index files decide the metric,
 * data_source.py:71; its wikipedia merge keeps the LOWEST scores, rerank.py:30, i.e. treats scores as distances).  Returns
 * the k nearest rows, nearest first, d_D = |q - x|^2, ties by ascending id, padding (+inf, -1).
 *   d_half_sqnorm  device f32 [n_rows]: |x|^2 / 2 of every stored row (rr_half_sqnorms). */
int rr_flat_search_l2(const void* d_xb, const float* d_half_sqnorm, int dtype, int64_t n_rows, int dim, const void* d_xq,
                      int nq, int k, float* d_D, int64_t* d_I, int64_t id_offset, void* d_ws, size_t ws_bytes,
                      const uint8_t* d_route_mask, int64_t mask_stride, void* stream);
/* |x|^2 / 2 (f32) of every row of a stored corpus, computed from the stored (rounded) values. */
int rr_half_sqnorms(const void* d_xb, int dtype, int64_t n_rows, int dim, float* d_out, void* stream);

/* ---- Optional: the same exact inner-product top-k through an int8 screening copy --------------------------------------
 * Same call site as rr_flat_search (`index.search`, ragroute/data_source.py:158,186,203) and the same results; the
 * reference has no counterpart (faiss.IndexFlatIP reads every f32 row).  The corpus keeps its f16/bf16 rows and one int8
 * copy beside them (+50 % HBM); a search streams only the int8 copy (half the bytes, twice the MFMA rate) to rank rows by
 * the integer dot product, keeps the list_len best per query, re-scores those from the f16/bf16 rows in f32 and checks a
 * rigorous quantisation-error bound: d_exact[q] = 1 means the returned k rows are proven to be the exact top-k of the
 * f16/bf16 corpus; 0 means the proof failed for that query (list_len too short for this data) and the caller must repeat the
 * batch with rr_flat_search.  Inner product / cosine only; dim <= 1536.
 *   rr_screen_dim(dim)   bytes per int8 row for rows of dim f16/bf16 elements (next multiple of 256), <0 if unsupported
 *   rr_screen_build      d_x8 device int8 [n_rows][rr_screen_dim(dim)] and d_stats device f32[8] (corpus scale and error
 *                        maxima) from the stored rows; rebuild after the corpus changes
 *   rr_flat_search_screened  arguments as rr_flat_search, plus d_x8 / d_stats, list_len in [k, 1024] and d_exact u8 [nq]. */
int rr_screen_dim(int dim);
int rr_screen_build(const void* d_xb, int dtype, int64_t n_rows, int dim, void* d_x8, float* d_stats, void* stream);
size_t rr_flat_search_screened_workspace_bytes(int k, int list_len, int nq, int dim);
int rr_flat_search_screened(const void* d_xb, int dtype, const void* d_x8, const float* d_stats, int64_t n_rows, int dim,
                            const void* d_xq, int nq, int k, int list_len, float* d_D, int64_t* d_I, int64_t id_offset,
                            uint8_t* d_exact, void* d_ws, size_t ws_bytes, const uint8_t* d_route_mask, int64_t mask_stride,
                            void* stream);

/* Measurement aid (bench.py): between rr_profile_begin and rr_profile_end every launch of the scan
 * kernel made by rr_flat_search on the calling thread is bracketed by HIP events on the launch stream.
 * rr_profile_end waits for them and returns the summed kernel time, the number of launches and the
 * corpus rows they covered.  No reference counterpart (the reference only has time.time() deltas,
 * data_source.py:104,130). */
int rr_profile_begin(int max_launches);
/* RR_ERR_WORKSPACE (after filling the outputs with what WAS recorded) if more than max_launches launches were made. */
int rr_profile_end(double* scan_ms_total, int* n_launches, double* rows_scanned);
/* Name of the scan kernel rr_flat_search dispatches the filter launches of a (dim, nq-query block) search to — the kernel
 * `roofline` figures refer to; "" if dim is not a padded dim. */
const char* rr_flat_scan_kernel_name(int dim, int nq);

/* Cross-source candidate merge: per query, the k best of m (score, id) candidates.
 * Replaces `np.argsort(scores)[::-1][:k]` / `np.argsort(scores)[:k]` — reference
 * ragroute/rerank.py:3-9 (rerank_medrag) and :28-34 (rerank_wikipedia) — applied to the
 * concatenated per-source lists of ragroute/http_server.py:280-293.  descending != 0 keeps the
 * highest scores; ties by ascending id; entries with id < 0 are padding and sort last.
 *   d_Din f32 [nq][m], d_Iin i64 [nq][m]  ->  d_Dout f32 [nq][k], d_Iout i64 [nq][k]. m <= 8192. */
int rr_merge_topk(const float* d_Din, const int64_t* d_Iin, int nq, int m, int k, int descending,
                  float* d_Dout, int64_t* d_Iout, void* stream);

/* The same merge reading the candidate exchange buffer in place, as the all-gather left it (no repacking copies):
 * n_ranks blocks, rank_stride_bytes apart, each `D f32 [slots][nq][k_in]` at offset 0 and `I i64 [slots][nq][k_in]` at
 * ids_offset_bytes (one slot per data source the rank holds; unused slots are padding (-inf, -1)).  Candidate order per
 * query = rank-major, then slot, then the source's own best-first order: the concatenation the reference's front-end
 * builds from the per-source replies (ragroute/http_server.py:280-286) before rerank.py:3-9.
 * n_ranks * slots * k_in <= 8192; both byte counts multiples of 8. */
int rr_merge_topk_gathered(const void* d_gathered, int n_ranks, size_t rank_stride_bytes, size_t ids_offset_bytes, int slots,
                           int nq, int k_in, int k, int descending, float* d_Dout, int64_t* d_Iout, void* stream);

/* Router MLP (CorpusRoutingNN) with the feature build and StandardScaler folded into fc1.
 * Replaces, for a batch of queries, reference ragroute/router.py:241-283
 * (select_relevant_sources_ragroute: pad/concat features, scaler.transform, model forward,
 * sigmoid, threshold) and router.py:50-55 (CorpusRoutingNN.forward).  All pointers device f32. */
typedef struct rr_router_weights {
  int32_t n_sources;               /* C: rows per query */
  int32_t d_max;                   /* padded query-embedding length (config.py:92-96) */
  int32_t n_models;                /* distinct query-embedding models M */
  int32_t reserved;
  const int32_t* model_of_source;  /* [C] which of the M embeddings source c uses */
  const float* w1q;                /* [d_max][256] fc1 query block, transposed, scaler folded */
  const float* c1;                 /* [C][256] per-source constant: b1 + centroid + one-hot + scaler mean terms */
  const float* ln1_g; const float* ln1_b;   /* [256] */
  const float* w2;                 /* [256][128] fc2 transposed */
  const float* b2; const float* ln2_g; const float* ln2_b; /* [128] */
  const float* w3;                 /* [128] */
  float b3;
  float prob_threshold;            /* 0.4924 medrag, 0.5 otherwise (router.py:277-280) */
  float ln_eps;                    /* 1e-5 */
  float reserved2;
} rr_router_weights;
/*   d_xq     f32 [nq][M][d_max] zero-padded query embeddings (one per model)
 *   d_logits f32 [nq][C]   fc3 output
 *   d_mask   u8  [nq][C]   1 where sigmoid(logit) > prob_threshold */
int rr_router_mlp(const rr_router_weights* w, const float* d_xq, int nq, float* d_logits, uint8_t* d_mask,
                  void* stream);
/* The same operator for BATCHES (what the batched router service, the pipeline and bench.py call): from 32 queries on, fc1
 * and fc2 run on the f32 matrix cores with 64 queries sharing every weight load; partial sums of fc1 live in a caller-provided
 * device workspace of rr_router_workspace_bytes(w, nq) bytes (0 for batches that take the small-batch kernel above, for which
 * d_ws may be NULL).  The reference has no batched form: it forwards one query at a time (router.py:207-219, 241-283). */
size_t rr_router_workspace_bytes(const rr_router_weights* w, int nq);
int rr_router_mlp_ws(const rr_router_weights* w, const float* d_xq, int nq, float* d_logits, uint8_t* d_mask, void* d_ws,
                     size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAGROUTE_HIP_H */
