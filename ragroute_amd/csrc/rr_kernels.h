// Internal launch interface between the kernel translation units and the C-ABI (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ragroute_hip.h"

namespace rr {

constexpr int kMaxSegments = 32;   // RR_MAX_SEGMENTS
constexpr int kMaxChunks = 12;     // scan launches of one search after the bootstrap (chunk_schedule picks <= 8)

// One contiguous run of tiles inside a chunk launch of a segmented search: launch ordinals [previous j_end, j_end) map to tiles
// j + delta, all in segment `seg` whose valid rows end at row_limit.  A launch has one entry per segment with a non-empty slice.
struct RangeEntry { uint32_t j_end; int32_t delta; uint32_t seg; uint32_t row_limit; };

// Segmented search (rr_flat_search_segments): one matrix holds several data sources that receive the same query block.  The
// scan walks the whole matrix as one corpus of "virtual" rows; these device tables (written by prep_kernel) say which rows
// belong to which source.  Segment s owns tiles [tile_end[s-1], tile_end[s]) (its alignment gap included) and its VALID rows
// are [row_begin[s], row_limit[s]).
struct SegTables {
  const uint32_t* tile_end;    // [n] first tile past segment s
  const uint32_t* row_limit;   // [n] first row past the valid rows of segment s
  const uint32_t* row_begin;   // [n]
  const int64_t* id_offset;    // [n] result id = id_offset[s] + (row - row_begin[s])
  const uint32_t* sel;         // [256] per query: bit s set = the router selected segment s for it
  uint32_t n;                  // 0 = plain search (one corpus, no tables)
};

struct ScanArgs {
  const void* xb;       // [n_rows][D] f16/bf16 corpus (D = padded dim)
  const void* xq;       // [nq][D] queries, same dtype
  const void* xqs;      // the same queries in MFMA-fragment order (prep kernel): [wave 4][block 4 or 2][k slice D/32][lane 64][8 elements] (wide rows: 16 blocks)
  const float* thr;     // [256] strict thresholds (filter mode); segmented: [n_segs][256], +inf where the query is not routed to the segment
  const RangeEntry* ranges;   // segmented search, chunk launches: the launch's tile runs (tile_first = 0, tile_stride = 1); nullptr otherwise
  uint32_t n_ranges;
  uint32_t ties_pass;         // segmented search: rows are not scanned in ascending id order, so a refreshed threshold must let ties pass
  uint64_t* cand;       // [256][grid*2][cap] candidate keys (filter mode)
  uint32_t* cand_cnt;   // [256][grid*2]
  uint64_t* scratch;    // [grid*4][cap] per-wave compaction scratch
  float* dense;         // [256][dense_ld] scores (dense mode)
  const float* half_sqnorm;  // L2 metric: |x|^2/2 per row, else nullptr
  uint64_t* timeline;        // diagnostics (RR_SCAN_TIMELINE): per-workgroup start / end s_memrealtime, else nullptr
  uint32_t n_rows, nq;
  uint32_t tile_first, tile_stride, n_tiles;  // tile(j) = tile_first + j*tile_stride, j < n_tiles
  uint32_t dense_ld;
  int cap, k;
};

// flat_scan.hip
hipError_t launch_flat_scan(const ScanArgs& a, int dtype, int D, bool dense, int grid, hipStream_t st);
int scan_padded_dim(int d);
const char* scan_kernel_name(int D, int nq, bool l2);  // filter-launch kernel for a block of nq queries
int scan_bufs_per_wg(int D, int nq, bool l2, int k);   // candidate buffers per (workgroup, query) of the kernel that will serve the block
int scan_queries_per_launch(int D, int nq);  // nq = queries of the whole call
int scan_query_blocks_per_wave(int D, int nq, bool l2);  // 4 / 2: the kernel reads the fragment-order copy; 0: row-major queries

// select.hip
struct SelectArgs {
  float* thr;           // [256]; segmented: [n_segs][256]
  SegTables seg;        // seg.n == 0: plain search
  uint64_t* list;       // [256][list_ld] running top-k keys, sorted descending
  uint32_t* list_cnt;   // [256]
  const uint64_t* cand; const uint32_t* cand_cnt;  // as ScanArgs
  const float* dense; uint32_t dense_ld, dense_cols;
  uint32_t tile_first, tile_stride;  // dense column c <-> row (tile_first + (c/32)*tile_stride)*32 + c%32
  uint32_t n_rows, nq;
  uint32_t nbuf, list_ld;
  int cap, k;
};
// init_state + copy of the query block into fragment order (xqs; nullptr or blocks_per_wave == 0: init only)
// segs != nullptr: also writes the device segment tables (a.seg points at them) and the per-query selection bits from the route mask
struct SegHost {                      // by value in prep_kernel's kernarg
  uint32_t n;
  uint32_t n_chunks;                  // chunk launches after the bootstrap, and the cumulative fraction each one reaches
  uint32_t frac[kMaxChunks];
  RangeEntry* ranges;                 // device [kMaxChunks][kMaxSegments], written by prep_kernel
  uint32_t tile_end[kMaxSegments], row_limit[kMaxSegments], row_begin[kMaxSegments];
  int32_t mask_col[kMaxSegments];
  int64_t id_offset[kMaxSegments];
};
hipError_t launch_prep(const SelectArgs& a, const void* xq, void* xqs, int dim, int blocks_per_wave, hipStream_t st,
                       const SegHost* segs = nullptr, const uint8_t* route_mask = nullptr, int64_t mask_stride = 0);
hipError_t launch_dense_select(const SelectArgs& a, bool bootstrap, hipStream_t st);
// fin != nullptr: this is the last compaction of an inner-product search, emit (D, I) directly (no finalize launch)
struct FinalizeArgs { float* D; int64_t* I; int64_t id_offset; const uint8_t* mask; int64_t mask_stride; };
hipError_t launch_compact(const SelectArgs& a, const FinalizeArgs* fin, hipStream_t st);
hipError_t launch_finalize(const SelectArgs& a, float* D, int64_t* I, int64_t id_offset, const uint8_t* mask, int64_t mask_stride,
                           const void* xq_l2, int dtype, int dim, hipStream_t st);  // xq_l2 != null: emit squared L2 distances
// Where the merge reads its candidates: n_outer x n_inner lists of [nq][k_in] (score, id) entries; list (o, i) starts at
// D + o * outer_D + i * inner (elements), ids likewise with outer_I.  Plain [nq][m] input: one list, k_in = m.
struct MergeSrc { const float* D; const int64_t* I; int k_in, n_inner; int64_t outer_D, outer_I, inner; };
hipError_t launch_merge_topk(const MergeSrc& src, int nq, int m, int k, int descending, float* Dout, int64_t* Iout, hipStream_t st);

// prep.hip
hipError_t launch_l2_normalize_f32(float* x, int64_t n, int64_t d, hipStream_t st);
hipError_t launch_rows_to_half(const float* x, int64_t n, int64_t d, int64_t ld_in, void* out, int dtype, int64_t d_out,
                               int normalize, hipStream_t st);

hipError_t launch_half_sqnorms(const void* xb, int dtype, int64_t n, int dim, float* out, hipStream_t st);
hipError_t launch_centroid(const void* xb, int dtype, int64_t n, int dim, int d, float* out, hipStream_t st);

// screen.hip (int8 screening copy + exact re-scoring)
hipError_t launch_screen_build(const void* xb, int dtype, int64_t n, int dim, int8_t* x8, int dim8, uint32_t* stats, hipStream_t st);
hipError_t launch_screen_queries(const void* xq, int dtype, int nq, int dim, int8_t* q8, int dim8, const uint32_t* stats, float* qinfo,
                                 hipStream_t st);
hipError_t launch_rescore(const void* xb, const void* xq, int dtype, int dim, int nq, int L, const float* PL, const int64_t* IL,
                          const float* qinfo, int k, float* D, int64_t* I, int64_t id_offset, uint8_t* exact, const uint8_t* mask,
                          int64_t mask_stride, hipStream_t st);

// router.hip
hipError_t launch_router_mlp(const rr_router_weights* w, const float* xq, int nq, float* logits, uint8_t* mask,
                             hipStream_t st);
constexpr int kRouterMfmaMinQueries = 32;   // batches from this size take the matrix-core form (needs a workspace)
size_t router_workspace_bytes(const rr_router_weights* w, int nq);
hipError_t launch_router_mlp_ws(const rr_router_weights* w, const float* xq, int nq, float* logits, uint8_t* mask, void* ws,
                                hipStream_t st);

}  // namespace rr
