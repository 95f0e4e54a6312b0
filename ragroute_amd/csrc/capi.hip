// C ABI of libragroute_hip.so (declared in include/ragroute_hip.h) and the host-side schedule of
// the flat search: bootstrap threshold -> geometrically growing scan chunks, each followed by an
// exact compaction that publishes the next per-query threshold -> finalize.  Everything is
// enqueued on the caller's stream; nothing here synchronises with the device.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "rr_common.h"
#include "rr_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* detail = "") {
  snprintf(g_err, sizeof(g_err), fmt, detail);
  return code;
}
int hip_fail(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return RR_ERR_HIP;
}

int device_cus() {
  static thread_local int cached_dev = -1, cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (dev != cached_dev) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    cached = v;
    cached_dev = dev;
  }
  return cached;
}

// ---- live profiling of the dominant kernel (bench.py roofline) -------------------------------------
struct ProfEntry { hipEvent_t start, stop; uint64_t rows; };
thread_local ProfEntry* g_prof = nullptr;
thread_local int g_prof_cap = 0, g_prof_n = 0, g_prof_dropped = 0;

hipError_t profiled_scan(const rr::ScanArgs& a, int dtype, int D, bool dense, int grid, hipStream_t st) {
  if (g_prof && g_prof_n >= g_prof_cap) ++g_prof_dropped;   // rr_profile_end reports it: a silent drop would overstate the roofline
  if (g_prof && g_prof_n < g_prof_cap) {
    ProfEntry& p = g_prof[g_prof_n++];
    p.rows = (uint64_t)a.n_tiles * rr::kTileRows;
    (void)hipEventRecord(p.start, st);
    hipError_t e = rr::launch_flat_scan(a, dtype, D, dense, grid, st);
    (void)hipEventRecord(p.stop, st);
    return e;
  }
  return rr::launch_flat_scan(a, dtype, D, dense, grid, st);
}

// Schedule knobs (defaults from rr_common.h; RR_SAMPLE_ROWS / RR_CHUNK_GROWTH override them for tuning runs)
int g_sample_rows = rr::kSampleRows, g_chunk_growth = 0;  // 0 = size-aware schedule
void read_schedule_env() {
  static const bool once = [] {
    if (const char* v = rr::tuning_env("RR_SAMPLE_ROWS")) {
      int r = atoi(v) / rr::kTileRows * rr::kTileRows;
      if (r >= 1024 && r <= rr::kSampleRows) g_sample_rows = r;
    }
    if (const char* v = rr::tuning_env("RR_CHUNK_GROWTH")) {
      int g = atoi(v);
      if (g >= 2 && g <= 1024) g_chunk_growth = g;
    }
    return true;
  }();
  (void)once;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Growth factor of the chunk schedule for a shard that is `ratio` times the bootstrap sample.  With c chunks the growth is
// g = ratio^(1/c) and every chunk leaves ~(g - 1) k survivors per query.  A workgroup sees every query, so it takes the
// insertion path ~(g - 1) k times per chunk, and its four waves wait for each other at the next tile barrier.  Cost model
// (MI355X, measured in round 2: a 2-chunk schedule at 10M rows ran 0.16 ms SLOWER than the 4-chunk one): ~37 us per chunk
// for the launch's fixed part and its compaction, ~0.18 us per insertion event of a workgroup.
// 1M x 768, k = 32 -> 3 chunks; 10M -> 4; 80M at k = 100 -> 6.
double chunk_schedule(double ratio, int k) {
  if (g_chunk_growth > 0) return (double)g_chunk_growth;   // RR_CHUNK_GROWTH pins the round-1 schedule for A/B runs
  if (ratio <= 1.0) return 2.0;
  double best_cost = 1e30, best_g = 8.0;
  for (int c = 1; c <= 8; ++c) {
    const double g = pow(ratio, 1.0 / c);
    const double cost = c * (37.0 + 0.18 * (g - 1.0) * k);
    if (cost < best_cost) { best_cost = cost; best_g = g; }
  }
  return best_g < 1.5 ? 1.5 : best_g;
}

struct Workspace {
  uint32_t *seg_tile_end, *seg_row_limit, *seg_row_begin, *seg_sel;   // segmented search: device tables written by prep_kernel
  int64_t* seg_id_offset;
  rr::RangeEntry* ranges;                                             // [kMaxChunks][kMaxSegments] tile runs of the chunk launches
  float* thr;
  uint32_t* list_cnt;
  uint64_t* list;
  uint32_t* cand_cnt;
  uint64_t* scratch;
  float* dense;
  uint64_t* cand;
  void* xqs;
  size_t total;
};

Workspace carve(char* base, int k, int grid, int dim) {
  using namespace rr;
  const int cap = cand_cap_for_k(k);
  // candidate buffers per (workgroup, query): 4 lane quarters; the row-split wide-row kernel (dim > 768) uses 8
  const int bpw = dim > kMaxResidentDim ? 8 : 4;
  Workspace w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return base ? base + o : (char*)nullptr; };
  w.thr = (float*)take((size_t)kMaxSegments * kQueriesPerBlock * sizeof(float));   // [segment][query] (plain search: row 0)
  w.seg_tile_end = (uint32_t*)take(kMaxSegments * sizeof(uint32_t));
  w.seg_row_limit = (uint32_t*)take(kMaxSegments * sizeof(uint32_t));
  w.seg_row_begin = (uint32_t*)take(kMaxSegments * sizeof(uint32_t));
  w.seg_id_offset = (int64_t*)take(kMaxSegments * sizeof(int64_t));
  w.seg_sel = (uint32_t*)take(kQueriesPerBlock * sizeof(uint32_t));
  w.ranges = (RangeEntry*)take((size_t)kMaxChunks * kMaxSegments * sizeof(RangeEntry));
  w.list_cnt = (uint32_t*)take(kQueriesPerBlock * sizeof(uint32_t));
  w.list = (uint64_t*)take((size_t)kQueriesPerBlock * kMaxK * sizeof(uint64_t));
  w.cand_cnt = (uint32_t*)take((size_t)kQueriesPerBlock * grid * bpw * sizeof(uint32_t));
  w.scratch = (uint64_t*)take((size_t)grid * 8 * cap * sizeof(uint64_t));  // up to 2 workgroups per CU x 4 waves
  w.dense = (float*)take((size_t)kQueriesPerBlock * kSampleRows * sizeof(float));
  w.cand = (uint64_t*)take((size_t)kQueriesPerBlock * grid * bpw * cap * sizeof(uint64_t));
  w.xqs = take((size_t)kQueriesPerBlock * kMaxDim * 2);  // the query block in MFMA-fragment order (prep kernel): 256 queries x the padded dim
  w.total = off;
  return w;
}

}  // namespace

// The compile flags of this library, written by ragroute_amd/_build.py next to the objects (its -I directory) before it compiles this unit.
#if __has_include("rr_build_flags.inc")
const char kBuildFlags[] =
#include "rr_build_flags.inc"
    ;
#else
const char kBuildFlags[] = "unknown (not built by ragroute_amd/_build.py)";
#endif

extern "C" {

int rr_version(void) { return 500; }  // 0.5.0: row-split wide-row kernel for L2 / any k / 193+ queries (8 candidate buffers per workgroup on wide rows at every k: larger workspace), tuning variables ignored by product builds, cursor-free plain scans
const char* rr_build_flags(void) { return kBuildFlags; }
const char* rr_last_error(void) { return g_err; }
int rr_device_cus(void) {
  int v = device_cus();
  return v > 0 ? v : fail(RR_ERR_HIP, "rr_device_cus: no HIP device%s");
}
int rr_padded_dim(int d) {
  if (d <= 0) return RR_ERR_INVALID;
  int p = rr::scan_padded_dim(d);
  return p > 0 ? p : RR_ERR_UNSUPPORTED;
}

int rr_l2_normalize_f32(float* x, int64_t n, int64_t d, void* stream) {
  if (n < 0 || d <= 0 || (!x && n > 0)) return fail(RR_ERR_INVALID, "rr_l2_normalize_f32: bad arguments%s");
  hipError_t e = rr::launch_l2_normalize_f32(x, n, d, (hipStream_t)stream);
  return e == hipSuccess ? RR_OK : hip_fail(e, "rr_l2_normalize_f32");
}

int rr_rows_to_half(const float* x, int64_t n, int64_t d, int64_t ld_in, void* out, int dtype, int64_t d_out,
                    int normalize, void* stream) {
  if (n < 0 || d <= 0 || ld_in < d || d_out < d || ((!x || !out) && n > 0))
    return fail(RR_ERR_INVALID, "rr_rows_to_half: bad arguments%s");
  if (dtype != RR_DTYPE_F16 && dtype != RR_DTYPE_BF16) return fail(RR_ERR_INVALID, "rr_rows_to_half: bad dtype%s");
  hipError_t e = rr::launch_rows_to_half(x, n, d, ld_in, out, dtype, d_out, normalize, (hipStream_t)stream);
  return e == hipSuccess ? RR_OK : hip_fail(e, "rr_rows_to_half");
}

int rr_centroid(const void* xb, int dtype, int64_t n_rows, int dim, int d, float* out, void* stream) {
  if (n_rows < 0 || d < 1 || d > dim || dim % 64 != 0 || !out || (!xb && n_rows > 0))
    return fail(RR_ERR_INVALID, "rr_centroid: bad arguments%s");
  if (dtype != RR_DTYPE_F16 && dtype != RR_DTYPE_BF16) return fail(RR_ERR_INVALID, "rr_centroid: bad dtype%s");
  hipError_t e = rr::launch_centroid(xb, dtype, n_rows, dim, d, out, (hipStream_t)stream);
  return e == hipSuccess ? RR_OK : hip_fail(e, "rr_centroid");
}

size_t rr_flat_search_workspace_bytes(int k) {   // any dim
  if (k < 1 || k > rr::kMaxK) return 0;
  int grid = device_cus();
  if (grid <= 0) return 0;
  return carve(nullptr, k, grid, rr::kMaxDim).total;
}
size_t rr_flat_search_workspace_bytes_for(int k, int dim) {
  if (k < 1 || k > rr::kMaxK || rr::scan_padded_dim(dim) != dim) return 0;
  int grid = device_cus();
  if (grid <= 0) return 0;
  return carve(nullptr, k, grid, dim).total;
}

// segs != nullptr: segmented search — xb is ONE matrix of n_rows rows holding several sources (segs->row_begin / row_limit),
// scanned as one corpus; route_mask then is [nq][mask_stride] with one column per segment (segs->mask_col), id_offset unused.
static int flat_search_impl(const void* xb, int dtype, int64_t n_rows, int dim, const void* xq, int nq, int k, float* D,
                            int64_t* I, int64_t id_offset, void* ws, size_t ws_bytes, const uint8_t* route_mask,
                            int64_t mask_stride, const float* half_sqnorm, void* stream, rr::SegHost* segs = nullptr) {
  using namespace rr;
  hipStream_t st = (hipStream_t)stream;
  if (k < 1 || k > kMaxK) return fail(RR_ERR_INVALID, "rr_flat_search: k must be in [1, 1024]%s");
  if (nq < 0 || n_rows < 0) return fail(RR_ERR_INVALID, "rr_flat_search: negative size%s");
  if (n_rows > 0xFFFFFFE0ll) return fail(RR_ERR_UNSUPPORTED, "rr_flat_search: more than 2^32-32 rows per shard%s");
  if (dtype != RR_DTYPE_F16 && dtype != RR_DTYPE_BF16 && dtype != kDtypeI8) return fail(RR_ERR_INVALID, "rr_flat_search: bad dtype%s");
  if (scan_padded_dim(dim) != dim)
    return fail(RR_ERR_UNSUPPORTED, "rr_flat_search: dim must equal rr_padded_dim(d)%s");
  if (nq == 0) return RR_OK;
  if (!xq || !D || !I || !ws || (!xb && n_rows > 0)) return fail(RR_ERR_INVALID, "rr_flat_search: null pointer%s");
  const int grid = device_cus();
  if (grid <= 0) return fail(RR_ERR_HIP, "rr_flat_search: no HIP device%s");
  Workspace w = carve((char*)ws, k, grid, dtype == kDtypeI8 ? kMaxResidentDim : dim);
  if (ws_bytes < w.total) return fail(RR_ERR_WORKSPACE, "rr_flat_search: workspace smaller than rr_flat_search_workspace_bytes_for(k, dim)%s");

  read_schedule_env();
  const int cap = cand_cap_for_k(k);
  const uint32_t total_tiles = (uint32_t)((n_rows + kTileRows - 1) / kTileRows);
  hipError_t e;
#define RR_CHECK(call, what) do { e = (call); if (e != hipSuccess) return hip_fail(e, what); } while (0)

  if (segs && n_rows > kDenseMaxRows) {
    // chunk fractions of the segmented schedule: frac[c] = sample share x growth^(c+1), the last one 1 (no sliver of a last chunk)
    const uint32_t n_sample_tiles = (uint32_t)g_sample_rows / kTileRows;
    const double ratio = (double)total_tiles / n_sample_tiles, growth = chunk_schedule(ratio, k);
    double f = growth / ratio;
    uint32_t c = 0;
    for (; c + 1 < (uint32_t)kMaxChunks && f * 1.25 < 1.0; ++c, f *= growth) segs->frac[c] = (uint32_t)(f * 4294967296.0);
    segs->frac[c++] = 0xFFFFFFFFu;
    segs->n_chunks = c;
    segs->ranges = w.ranges;
  }
  const int qpl = scan_queries_per_launch(dim, nq);  // 256, or 128 where only 32 queries per wave stay resident and the batch is small
  for (int qb = 0; qb < nq; qb += qpl) {
    const int nqb = nq - qb < qpl ? nq - qb : qpl;
    const char* xq_b = (const char*)xq + (size_t)qb * dim * 2;
    float* D_b = D + (size_t)qb * k;
    int64_t* I_b = I + (size_t)qb * k;

    SelectArgs s;
    memset(&s, 0, sizeof(s));
    s.thr = w.thr; s.list = w.list; s.list_cnt = w.list_cnt; s.cand = w.cand; s.cand_cnt = w.cand_cnt;
    s.dense = w.dense; s.dense_ld = kSampleRows; s.n_rows = (uint32_t)n_rows; s.nq = (uint32_t)nqb;
    s.nbuf = (uint32_t)grid * (uint32_t)(dtype == kDtypeI8 ? 4 : scan_bufs_per_wg(dim, nqb, half_sqnorm != nullptr, k)); s.list_ld = kMaxK; s.cap = cap; s.k = k;
    if (segs) s.seg = SegTables{w.seg_tile_end, w.seg_row_limit, w.seg_row_begin, w.seg_id_offset, w.seg_sel, segs->n};
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.xb = xb; a.xq = xq_b; a.thr = w.thr; a.cand = w.cand; a.cand_cnt = w.cand_cnt; a.scratch = w.scratch;
    a.dense = w.dense; a.n_rows = (uint32_t)n_rows; a.nq = (uint32_t)nqb; a.dense_ld = kSampleRows; a.cap = cap; a.k = k;
    a.half_sqnorm = half_sqnorm;
    if (segs) a.ties_pass = 1;
    // (per query block: launch_flat_scan picks the kernel from the block's own query count, e.g. the 44-query tail of a 300-query call)
    const int bpw = dtype == kDtypeI8 ? 4 : scan_query_blocks_per_wave(dim, nqb, half_sqnorm != nullptr);
    a.xqs = bpw ? w.xqs : nullptr;

    bool finalized = false;
    // (segmented: the per-query route mask rows of this query block; folded into the thresholds, not into the finalize)
    const uint8_t* mask_b = route_mask ? route_mask + (size_t)qb * mask_stride : nullptr;
    RR_CHECK(launch_prep(s, xq_b, w.xqs, dim, bpw, st, segs, segs ? mask_b : nullptr, mask_stride), "rr_flat_search/prep");
    if (n_rows > 0 && n_rows <= kDenseMaxRows) {
      // tiny corpus: all scores, one exact selection
      a.tile_first = 0; a.tile_stride = 1; a.n_tiles = total_tiles;
      RR_CHECK(profiled_scan(a, dtype, dim, true, grid, st), "rr_flat_search/dense");
      s.tile_first = 0; s.tile_stride = 1; s.dense_cols = total_tiles * kTileRows;
      RR_CHECK(launch_dense_select(s, false, st), "rr_flat_search/dense_select");
    } else if (n_rows > 0) {
      // bootstrap: k-th best score of a strided sample of tiles = a valid lower bound
      const uint32_t n_sample_tiles = (uint32_t)g_sample_rows / kTileRows;
      a.tile_first = 0; a.tile_stride = total_tiles / n_sample_tiles; a.n_tiles = n_sample_tiles;
      RR_CHECK(profiled_scan(a, dtype, dim, true, grid, st), "rr_flat_search/bootstrap");
      s.tile_first = 0; s.tile_stride = a.tile_stride; s.dense_cols = (uint32_t)g_sample_rows;
      RR_CHECK(launch_dense_select(s, true, st), "rr_flat_search/bootstrap_select");
      // chunks [0,e1), [e1,e2), ... with e growing geometrically: a chunk that is g times what came before it yields ~(g-1) k
      // survivors per query.  The number of chunks balances the fixed cost of a scan launch + compaction against the
      // insertion work of a long chunk (chunk_schedule).
      const double growth = chunk_schedule((double)total_tiles / n_sample_tiles, k);
      FinalizeArgs fin{D_b, I_b, id_offset, segs ? nullptr : mask_b, mask_stride};
      if (segs) {
        // Segmented: chunk c takes the next slice of EVERY segment (seg_cut), reaching the cumulative fraction frac[c] of each, so
        // that whatever subset of the segments a query is routed to, its rows arrive in geometrically growing portions.  The
        // launch sizes below and prep_kernel's range tables (already enqueued above with the same SegHost) use one expression.
        for (uint32_t c = 0; c < segs->n_chunks; ++c) {
          uint32_t n_launch_tiles = 0, n_ranges = 0;
          for (uint32_t sg = 0; sg < segs->n; ++sg) {
            const uint32_t tiles = (segs->row_limit[sg] - segs->row_begin[sg] + kTileRows - 1) / kTileRows;
            const uint32_t lo = c ? seg_cut(tiles, segs->frac[c - 1]) : 0u, hi = seg_cut(tiles, segs->frac[c]);
            if (hi > lo) { n_launch_tiles += (hi - lo + 7u) & ~7u; ++n_ranges; }
          }
          const bool last = c + 1 == segs->n_chunks;
          if (n_launch_tiles == 0 && !last) continue;
          if (n_launch_tiles > 0) {
            a.tile_first = 0; a.tile_stride = 1; a.n_tiles = n_launch_tiles;
            a.ranges = w.ranges + (size_t)c * kMaxSegments; a.n_ranges = n_ranges;
            RR_CHECK(profiled_scan(a, dtype, dim, false, grid, st), "rr_flat_search_segments/scan");
          }
          RR_CHECK(launch_compact(s, last ? &fin : nullptr, st), "rr_flat_search_segments/compact");
          if (last) finalized = true;
        }
      } else {
      uint64_t begin = 0;
      double endf = (double)n_sample_tiles * growth;  // in tiles
      while (begin < total_tiles) {
        uint64_t end = ((uint64_t)(endf + 0.5) + 7) & ~7ull;   // whole 256-row groups: the wide-row kernels' K order is a function of the global group (flat_scan_wide.hip)
        if (end > total_tiles || (double)total_tiles < endf * 1.25) end = total_tiles;  // no sliver of a last chunk
        if (end <= begin) end = begin + 8;
        if (end > total_tiles) end = total_tiles;
        a.tile_first = (uint32_t)begin; a.tile_stride = 1; a.n_tiles = (uint32_t)(end - begin);
        a.timeline = rr::tuning_env("RR_SCAN_TIMELINE") ? (uint64_t*)w.dense : nullptr;  // the dense buffer is idle during chunk scans
        RR_CHECK(profiled_scan(a, dtype, dim, false, grid, st), "rr_flat_search/scan");
        const bool last = end == total_tiles;
        RR_CHECK(launch_compact(s, (last && !half_sqnorm) ? &fin : nullptr, st), "rr_flat_search/compact");
        if (last && !half_sqnorm) finalized = true;
        begin = end;
        endf *= growth;
      }
      }
    }
    if (!finalized)
      RR_CHECK(launch_finalize(s, D_b, I_b, id_offset, segs ? nullptr : mask_b, mask_stride,
                               half_sqnorm ? (const void*)xq_b : nullptr, dtype, dim, st),
               "rr_flat_search/finalize");
  }
#undef RR_CHECK
  return RR_OK;
}

int rr_flat_search(const void* xb, int dtype, int64_t n_rows, int dim, const void* xq, int nq, int k, float* D,
                   int64_t* I, int64_t id_offset, void* ws, size_t ws_bytes, const uint8_t* route_mask,
                   int64_t mask_stride, void* stream) {
  if (dtype != RR_DTYPE_F16 && dtype != RR_DTYPE_BF16) return fail(RR_ERR_INVALID, "rr_flat_search: bad dtype%s");
  return flat_search_impl(xb, dtype, n_rows, dim, xq, nq, k, D, I, id_offset, ws, ws_bytes, route_mask, mask_stride, nullptr, stream);
}

int rr_flat_search_l2(const void* xb, const float* half_sqnorm, int dtype, int64_t n_rows, int dim, const void* xq, int nq, int k,
                      float* D, int64_t* I, int64_t id_offset, void* ws, size_t ws_bytes, const uint8_t* route_mask,
                      int64_t mask_stride, void* stream) {
  if (!half_sqnorm && n_rows > 0) return fail(RR_ERR_INVALID, "rr_flat_search_l2: null half_sqnorm%s");
  // an empty index has no norms to pass; any non-null marker selects the L2 finalize (padding +inf) and is never dereferenced
  static const float kEmpty = 0.f;
  return flat_search_impl(xb, dtype, n_rows, dim, xq, nq, k, D, I, id_offset, ws, ws_bytes, route_mask, mask_stride,
                          half_sqnorm ? half_sqnorm : &kEmpty, stream);
}

int rr_flat_search_segments(const void* xb, int dtype, int64_t n_rows_total, int dim, const rr_segment* segs, int n_segs,
                            const void* xq, int nq, int k, float* D, int64_t* I, void* ws, size_t ws_bytes,
                            const uint8_t* route_mask, int64_t mask_stride, void* stream) {
  if (dtype != RR_DTYPE_F16 && dtype != RR_DTYPE_BF16) return fail(RR_ERR_INVALID, "rr_flat_search_segments: bad dtype%s");
  if (n_segs < 1 || n_segs > RR_MAX_SEGMENTS || !segs) return fail(RR_ERR_INVALID, "rr_flat_search_segments: need 1 .. 32 segments%s");
  if (n_rows_total < 0 || n_rows_total > 0xFFFFFFE0ll) return fail(RR_ERR_INVALID, "rr_flat_search_segments: bad total row count%s");
  rr::SegHost h;
  memset(&h, 0, sizeof(h));
  h.n = (uint32_t)n_segs;
  int64_t prev_end = 0, valid_rows = 0;
  int max_col = -1;
  for (int s = 0; s < n_segs; ++s) {
    const rr_segment& g = segs[s];
    // (row_begin <= n_rows_total is checked first, so the subtraction cannot overflow whatever n_rows is)
    if (g.n_rows < 0 || g.row_begin < prev_end || g.row_begin % RR_SEGMENT_ALIGN != 0 || g.row_begin > n_rows_total ||
        g.n_rows > n_rows_total - g.row_begin)
      return fail(RR_ERR_INVALID, "rr_flat_search_segments: segments must be ascending, non-overlapping, inside the matrix, and begin at "
                                  "multiples of 256 rows%s");
    if (g.mask_col < -1) return fail(RR_ERR_INVALID, "rr_flat_search_segments: mask_col must be >= -1%s");
    prev_end = g.row_begin + g.n_rows;
    valid_rows += g.n_rows;
    if (g.mask_col > max_col) max_col = g.mask_col;
    h.row_begin[s] = (uint32_t)g.row_begin;
    h.row_limit[s] = (uint32_t)prev_end;
    h.mask_col[s] = route_mask ? g.mask_col : -1;
    h.id_offset[s] = g.id_offset;
  }
  if (route_mask && max_col >= 0 && mask_stride <= max_col)
    return fail(RR_ERR_INVALID, "rr_flat_search_segments: mask_stride must exceed every mask_col%s");
  // segment s owns the tiles up to the next segment's first row (its alignment gap included); the last one runs to the end of
  // the rows that are scanned at all: nothing behind the last valid row is read
  const int64_t scanned = prev_end;
  for (int s = 0; s < n_segs; ++s) {
    const int64_t next = s + 1 < n_segs ? segs[s + 1].row_begin : scanned;
    h.tile_end[s] = (uint32_t)((next + rr::kTileRows - 1) / rr::kTileRows);
  }
  (void)valid_rows;
  return flat_search_impl(xb, dtype, scanned, dim, xq, nq, k, D, I, 0, ws, ws_bytes, route_mask, mask_stride, nullptr, stream, &h);
}

// ---- int8 screening copy ---------------------------------------------------------------------------------
struct ScreenWs { int8_t* q8; float* qinfo; float* PL; int64_t* IL; char* inner; size_t inner_bytes, total; };
static ScreenWs carve_screen(char* base, int list_len, int nq, int dim8, int grid) {
  ScreenWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return base ? base + o : (char*)nullptr; };
  w.q8 = (int8_t*)take((size_t)nq * dim8);
  w.qinfo = (float*)take((size_t)nq * 2 * sizeof(float));
  w.PL = (float*)take((size_t)nq * list_len * sizeof(float));
  w.IL = (int64_t*)take((size_t)nq * list_len * sizeof(int64_t));
  w.inner_bytes = carve(nullptr, list_len, grid, rr::kMaxResidentDim).total;
  w.inner = take(w.inner_bytes);
  w.total = off;
  return w;
}

int rr_screen_dim(int dim) {
  if (dim < 8 || dim % 8 != 0) return RR_ERR_INVALID;
  const int d8 = (dim + 255) / 256 * 256;
  return d8 <= 2 * rr::kMaxResidentDim ? d8 : RR_ERR_UNSUPPORTED;
}

int rr_screen_build(const void* xb, int dtype, int64_t n_rows, int dim, void* x8, float* stats, void* stream) {
  const int dim8 = rr_screen_dim(dim);
  if (dim8 < 0) return fail(dim8, "rr_screen_build: the int8 screening copy needs dim <= 1536, a multiple of 8%s");
  if (n_rows < 0 || !stats || ((!xb || !x8) && n_rows > 0)) return fail(RR_ERR_INVALID, "rr_screen_build: bad arguments%s");
  if (dtype != RR_DTYPE_F16 && dtype != RR_DTYPE_BF16) return fail(RR_ERR_INVALID, "rr_screen_build: bad dtype%s");
  hipError_t e = rr::launch_screen_build(xb, dtype, n_rows, dim, (int8_t*)x8, dim8, (uint32_t*)stats, (hipStream_t)stream);
  return e == hipSuccess ? RR_OK : hip_fail(e, "rr_screen_build");
}

size_t rr_flat_search_screened_workspace_bytes(int k, int list_len, int nq, int dim) {
  const int dim8 = rr_screen_dim(dim);
  if (k < 1 || list_len < k || list_len > rr::kMaxK || nq < 0 || dim8 < 0) return 0;
  int grid = device_cus();
  if (grid <= 0) return 0;
  return carve_screen(nullptr, list_len, nq, dim8, grid).total;
}

int rr_flat_search_screened(const void* xb, int dtype, const void* x8, const float* stats, int64_t n_rows, int dim, const void* xq,
                            int nq, int k, int list_len, float* D, int64_t* I, int64_t id_offset, uint8_t* exact, void* ws,
                            size_t ws_bytes, const uint8_t* route_mask, int64_t mask_stride, void* stream) {
  using namespace rr;
  hipStream_t st = (hipStream_t)stream;
  const int dim8 = rr_screen_dim(dim);
  if (dim8 < 0) return fail(RR_ERR_UNSUPPORTED, "rr_flat_search_screened: dim must be a multiple of 8, <= 1536%s");
  if (k < 1 || list_len < k || list_len > kMaxK) return fail(RR_ERR_INVALID, "rr_flat_search_screened: need 1 <= k <= list_len <= 1024%s");
  if (dtype != RR_DTYPE_F16 && dtype != RR_DTYPE_BF16) return fail(RR_ERR_INVALID, "rr_flat_search_screened: bad dtype%s");
  if (nq < 0 || n_rows < 0) return fail(RR_ERR_INVALID, "rr_flat_search_screened: negative size%s");
  if (nq == 0) return RR_OK;
  if (!xq || !D || !I || !exact || !ws || !stats || ((!xb || !x8) && n_rows > 0))
    return fail(RR_ERR_INVALID, "rr_flat_search_screened: null pointer%s");
  const int grid = device_cus();
  if (grid <= 0) return fail(RR_ERR_HIP, "rr_flat_search_screened: no HIP device%s");
  ScreenWs w = carve_screen((char*)ws, list_len, nq, dim8, grid);
  if (ws_bytes < w.total) return fail(RR_ERR_WORKSPACE, "rr_flat_search_screened: workspace smaller than rr_flat_search_screened_workspace_bytes%s");
  hipError_t e = launch_screen_queries(xq, dtype, nq, dim, w.q8, dim8, (const uint32_t*)stats, w.qinfo, st);
  if (e != hipSuccess) return hip_fail(e, "rr_flat_search_screened/queries");
  // the int8 rows through the same scan + exact selection, k' = list_len; a row of dim8 bytes is dim8/2 two-byte elements
  int rc = flat_search_impl(x8, kDtypeI8, n_rows, dim8 / 2, w.q8, nq, list_len, w.PL, w.IL, 0, w.inner, w.inner_bytes, nullptr, 0, nullptr, stream);
  if (rc != RR_OK) return rc;
  e = launch_rescore(xb, xq, dtype, dim, nq, list_len, w.PL, w.IL, w.qinfo, k, D, I, id_offset, exact, route_mask, mask_stride, st);
  return e == hipSuccess ? RR_OK : hip_fail(e, "rr_flat_search_screened/rescore");
}

int rr_half_sqnorms(const void* xb, int dtype, int64_t n_rows, int dim, float* out, void* stream) {
  if (n_rows < 0 || dim < 8 || dim % 8 != 0 || ((!xb || !out) && n_rows > 0)) return fail(RR_ERR_INVALID, "rr_half_sqnorms: bad arguments%s");
  if (dtype != RR_DTYPE_F16 && dtype != RR_DTYPE_BF16) return fail(RR_ERR_INVALID, "rr_half_sqnorms: bad dtype%s");
  hipError_t e = rr::launch_half_sqnorms(xb, dtype, n_rows, dim, out, (hipStream_t)stream);
  return e == hipSuccess ? RR_OK : hip_fail(e, "rr_half_sqnorms");
}

int rr_profile_begin(int max_launches) {
  if (g_prof) return fail(RR_ERR_INVALID, "rr_profile_begin: already profiling%s");
  if (max_launches < 1) return fail(RR_ERR_INVALID, "rr_profile_begin: max_launches < 1%s");
  g_prof = new ProfEntry[max_launches];
  g_prof_cap = 0;
  g_prof_n = 0;
  g_prof_dropped = 0;
  for (int i = 0; i < max_launches; ++i) {
    if (hipEventCreate(&g_prof[i].start) != hipSuccess || hipEventCreate(&g_prof[i].stop) != hipSuccess)
      return fail(RR_ERR_HIP, "rr_profile_begin: hipEventCreate failed%s");
    g_prof_cap = i + 1;
  }
  return RR_OK;
}

int rr_profile_end(double* scan_ms_total, int* n_launches, double* rows_scanned) {
  if (!g_prof) return fail(RR_ERR_INVALID, "rr_profile_end: not profiling%s");
  double ms = 0, rows = 0;
  int n = 0;
  for (int i = 0; i < g_prof_n; ++i) {
    float t = 0;
    if (hipEventSynchronize(g_prof[i].stop) == hipSuccess && hipEventElapsedTime(&t, g_prof[i].start, g_prof[i].stop) == hipSuccess) {
      ms += t; rows += (double)g_prof[i].rows; ++n;
    }
  }
  for (int i = 0; i < g_prof_cap; ++i) { (void)hipEventDestroy(g_prof[i].start); (void)hipEventDestroy(g_prof[i].stop); }
  delete[] g_prof;
  g_prof = nullptr;
  g_prof_cap = g_prof_n = 0;
  if (scan_ms_total) *scan_ms_total = ms;
  if (n_launches) *n_launches = n;
  if (rows_scanned) *rows_scanned = rows;
  if (g_prof_dropped > 0) {
    char msg[32];
    snprintf(msg, sizeof(msg), "%d", g_prof_dropped);
    g_prof_dropped = 0;
    return fail(RR_ERR_WORKSPACE, "rr_profile_end: %s scan launches were not recorded (max_launches of rr_profile_begin too small); the totals returned are incomplete", msg);
  }
  return RR_OK;
}

const char* rr_flat_scan_kernel_name(int dim, int nq) {
  if (rr::scan_padded_dim(dim) != dim || nq < 1) return "";
  return rr::scan_kernel_name(dim, nq > rr::kQueriesPerBlock ? rr::kQueriesPerBlock : nq, false);
}

int rr_merge_topk(const float* Din, const int64_t* Iin, int nq, int m, int k, int descending, float* Dout,
                  int64_t* Iout, void* stream) {
  if (nq < 0 || m < 0 || k < 1) return fail(RR_ERR_INVALID, "rr_merge_topk: bad sizes%s");
  if (m > rr::kSelectCap) return fail(RR_ERR_UNSUPPORTED, "rr_merge_topk: m > 8192 candidates per query%s");
  if (nq == 0) return RR_OK;
  if ((m > 0 && (!Din || !Iin)) || !Dout || !Iout) return fail(RR_ERR_INVALID, "rr_merge_topk: null pointer%s");
  const rr::MergeSrc src{Din, Iin, m > 0 ? m : 1, 1, 0, 0, 0};
  hipError_t e = rr::launch_merge_topk(src, nq, m, k, descending, Dout, Iout, (hipStream_t)stream);
  return e == hipSuccess ? RR_OK : hip_fail(e, "rr_merge_topk");
}

int rr_merge_topk_gathered(const void* gathered, int n_ranks, size_t rank_stride_bytes, size_t ids_offset_bytes, int slots, int nq,
                           int k_in, int k, int descending, float* Dout, int64_t* Iout, void* stream) {
  if (n_ranks < 1 || slots < 1 || nq < 0 || k_in < 1 || k < 1) return fail(RR_ERR_INVALID, "rr_merge_topk_gathered: bad sizes%s");
  const size_t list_elems = (size_t)nq * k_in;
  if (rank_stride_bytes % 8 != 0 || ids_offset_bytes % 8 != 0 || ids_offset_bytes < (size_t)slots * list_elems * 4 ||
      rank_stride_bytes < ids_offset_bytes + (size_t)slots * list_elems * 8)
    return fail(RR_ERR_INVALID, "rr_merge_topk_gathered: the rank stride / id offset do not describe [D f32[slots][nq][k_in] | pad | I i64[slots][nq][k_in]]%s");
  const long long m = (long long)n_ranks * slots * k_in;
  if (m > rr::kSelectCap) return fail(RR_ERR_UNSUPPORTED, "rr_merge_topk_gathered: more than 8192 candidates per query%s");
  if (nq == 0) return RR_OK;
  if (!gathered || !Dout || !Iout) return fail(RR_ERR_INVALID, "rr_merge_topk_gathered: null pointer%s");
  if ((uintptr_t)gathered % 8 != 0) return fail(RR_ERR_INVALID, "rr_merge_topk_gathered: the buffer must be 8-byte aligned%s");
  const rr::MergeSrc src{(const float*)gathered, (const int64_t*)((const char*)gathered + ids_offset_bytes), k_in, slots,
                         (int64_t)(rank_stride_bytes / 4), (int64_t)(rank_stride_bytes / 8), (int64_t)list_elems};
  hipError_t e = rr::launch_merge_topk(src, nq, (int)m, k, descending, Dout, Iout, (hipStream_t)stream);
  return e == hipSuccess ? RR_OK : hip_fail(e, "rr_merge_topk_gathered");
}

int rr_router_mlp(const rr_router_weights* w, const float* xq, int nq, float* logits, uint8_t* mask, void* stream) {
  if (!w || nq < 0) return fail(RR_ERR_INVALID, "rr_router_mlp: bad arguments%s");
  if (w->n_sources < 1 || w->d_max < 1 || w->n_models < 1 || w->d_max > 8448)
    return fail(RR_ERR_INVALID, "rr_router_mlp: bad weight header%s");
  if (nq == 0) return RR_OK;
  if (!xq || !logits || !mask || !w->w1q || !w->c1 || !w->w2 || !w->w3 || !w->model_of_source)
    return fail(RR_ERR_INVALID, "rr_router_mlp: null pointer%s");
  hipError_t e = rr::launch_router_mlp(w, xq, nq, logits, mask, (hipStream_t)stream);
  return e == hipSuccess ? RR_OK : hip_fail(e, "rr_router_mlp");
}

size_t rr_router_workspace_bytes(const rr_router_weights* w, int nq) {
  if (!w || nq < 0 || w->n_sources < 1 || w->d_max < 1 || w->n_models < 1 || w->d_max > 8448) return 0;
  return rr::router_workspace_bytes(w, nq);
}

int rr_router_mlp_ws(const rr_router_weights* w, const float* xq, int nq, float* logits, uint8_t* mask, void* ws, size_t ws_bytes,
                     void* stream) {
  if (!w || nq < 0) return fail(RR_ERR_INVALID, "rr_router_mlp_ws: bad arguments%s");
  if (w->n_sources < 1 || w->d_max < 1 || w->n_models < 1 || w->d_max > 8448)
    return fail(RR_ERR_INVALID, "rr_router_mlp_ws: bad weight header%s");
  if (nq == 0) return RR_OK;
  if (!xq || !logits || !mask || !w->w1q || !w->c1 || !w->w2 || !w->w3 || !w->model_of_source)
    return fail(RR_ERR_INVALID, "rr_router_mlp_ws: null pointer%s");
  const size_t need = rr::router_workspace_bytes(w, nq);
  if (need == 0)  // small batch: the latency-oriented kernel, no workspace
    return rr_router_mlp(w, xq, nq, logits, mask, stream);
  if (!ws || ws_bytes < need) return fail(RR_ERR_WORKSPACE, "rr_router_mlp_ws: workspace smaller than rr_router_workspace_bytes%s");
  hipError_t e = rr::launch_router_mlp_ws(w, xq, nq, logits, mask, ws, (hipStream_t)stream);
  return e == hipSuccess ? RR_OK : hip_fail(e, "rr_router_mlp_ws");
}

}  // extern "C"
