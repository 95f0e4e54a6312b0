// K2 / K4: exact selection kernels around the scan.
//  * dense_select : top-k (or just the k-th score, for the bootstrap threshold) of a dense score row
//  * compact      : fold every workgroup's candidate buffers into the running per-query top-k and
//                   publish the next (strict) threshold
//  * finalize     : running list -> faiss-shaped (D f32[nq,k], I i64[nq,k])
//  * merge_topk   : cross-source (score, id) merge — reference ragroute/rerank.py:3-9, 28-34
// All of them order candidates by (score descending, id ascending); one workgroup per query,
// bitonic sort of 64-bit order keys in LDS.
#include <string.h>

#include "rr_common.h"
#include "rr_kernels.h"
#include "rr_sort.h"

namespace rr {

// ---- segmented search helpers ---------------------------------------------------------------------------------------------
// segment of a virtual row (<= 32 segments: a linear walk over the tile ends)
__device__ __forceinline__ int seg_of_row(const SegTables& t, uint32_t row) {
  const uint32_t tile = row / kTileRows;
  int s = 0;
  while (s + 1 < (int)t.n && tile >= t.tile_end[s]) ++s;
  return s;
}
// is `row` a row some source really holds (not alignment gap), in a segment the query is routed to?
__device__ __forceinline__ bool seg_row_counts(const SegTables& t, uint32_t row, uint32_t sel_bits) {
  const int s = seg_of_row(t, row);
  return row >= t.row_begin[s] && row < t.row_limit[s] && ((sel_bits >> s) & 1u);
}
__device__ __forceinline__ int64_t seg_result_id(const SegTables& t, uint32_t row) {
  const int s = seg_of_row(t, row);
  return t.id_offset[s] + (int64_t)(row - t.row_begin[s]);
}
// publish a query's threshold after a compaction (v = score of its k-th best so far): plain search one value, strict — every
// later row has a larger id, so a tie loses; segmented one per segment (+inf where the query is not routed), and ties PASS: the
// slices of the segments are scanned interleaved, so a later row can tie with a smaller id.
__device__ __forceinline__ void publish_thr(const SelectArgs& a, uint32_t q, float v) {
  if (a.seg.n == 0) { a.thr[q] = v; return; }
  const uint32_t bits = a.seg.sel[q];
  const float t = next_below(v);
  for (uint32_t s = 0; s < a.seg.n; ++s) a.thr[s * kQueriesPerBlock + q] = ((bits >> s) & 1u) ? t : __builtin_inff();
}

// One launch in front of every search: per-query state, and the query block copied into the order in which the query-resident scan
// kernels' waves load their MFMA B fragments: [wave][16-query block (4 per wave at d <= 768, 2 at 768 < d <= 1536)][32-wide k slice][lane][8 elements], so that each
// of the 96 loads of a wave (d = 768) is one contiguous KiB instead of 16 rows x 64 B.  (The row-major loads cost ~16 us of
// every scan launch: 5 launches per search.)  Slots past nq repeat the last query, as the kernel's clamped loads did.
// Segmented search (segs.n > 0): block 0 also writes the device segment tables and, per query, the selection bits (bit s = the
// route mask selects segment s for this query, or the segment has no mask column) and the initial thresholds per segment.
__global__ __launch_bounds__(256) void prep_kernel(SelectArgs a, const uint4* __restrict__ xq, uint4* __restrict__ xqs, int dim, int n_blocks,
                                                   SegHost segs, const uint8_t* __restrict__ route_mask, int64_t mask_stride) {
  if (blockIdx.x == 0) {
    const uint32_t q = threadIdx.x;
    a.list_cnt[q] = 0;
    if (segs.n == 0) {
      a.thr[q] = q < a.nq ? -__builtin_inff() : __builtin_inff();
    } else {
      if (q < segs.n) {
        const_cast<uint32_t*>(a.seg.tile_end)[q] = segs.tile_end[q];
        const_cast<uint32_t*>(a.seg.row_limit)[q] = segs.row_limit[q];
        const_cast<uint32_t*>(a.seg.row_begin)[q] = segs.row_begin[q];
        const_cast<int64_t*>(a.seg.id_offset)[q] = segs.id_offset[q];
      }
      if (q < segs.n_chunks) {   // thread c: the tile runs of chunk launch c (one per segment with a non-empty slice; see seg_cut)
        RangeEntry* out = segs.ranges + (size_t)q * kMaxSegments;
        RangeEntry* const first = out;
        uint32_t j = 0;
        for (uint32_t s = 0; s < segs.n; ++s) {
          const uint32_t base = segs.row_begin[s] / kTileRows;
          const uint32_t tiles = (segs.row_limit[s] - segs.row_begin[s] + kTileRows - 1) / kTileRows;
          const uint32_t lo = q ? seg_cut(tiles, segs.frac[q - 1]) : 0u, hi = seg_cut(tiles, segs.frac[q]);
          if (hi > lo) {
            const uint32_t len = (hi - lo + 7u) & ~7u;   // whole groups: the pad is the segment's alignment gap (rows >= row_limit)
            j += len;
            *out++ = RangeEntry{j, (int32_t)(base + lo) - (int32_t)(j - len), s, segs.row_limit[s]};
          }
        }
        if (out != first) out[-1].j_end = 0xFFFFFFFFu;   // the cursors stop at the last run without knowing how many there are
      }
      uint32_t bits = 0;
      if (q < a.nq)
        for (uint32_t s = 0; s < segs.n; ++s)
          if (segs.mask_col[s] < 0 || !route_mask || route_mask[(size_t)q * mask_stride + segs.mask_col[s]]) bits |= 1u << s;
      const_cast<uint32_t*>(a.seg.sel)[q] = bits;
      for (uint32_t s = 0; s < segs.n; ++s) a.thr[s * kQueriesPerBlock + q] = ((bits >> s) & 1u) ? -__builtin_inff() : __builtin_inff();
    }
  }
  if (!xqs) return;
  const int ks2 = dim >> 5;                       // 32-wide k slices
  const uint32_t total = (uint32_t)n_blocks * ks2 * 64u;   // 16-byte chunks: 16-query blocks x ks2 x 64 lanes
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
    const uint32_t lane = c & 63, s2 = (c >> 6) % ks2, wb = (c >> 6) / ks2;   // wb = wave * blocks per wave + block: queries 16 wb .. +15
    const uint32_t col = lane & 15, g = lane >> 4;
    uint32_t qi = wb * 16 + col;
    qi = qi < a.nq ? qi : a.nq - 1;
    xqs[c] = xq[((size_t)qi * dim + 32 * s2 + 8 * g) >> 3];
  }
}

// k-th largest 32-bit order key among the keys a 1024-thread workgroup holds in registers (PER each), by a 4-pass 8-bit
// radix select over LDS histograms.  Returns the key (0 if fewer than k non-zero keys exist).
template <int PER>
__device__ uint32_t radix_select_kth(const uint32_t (&key)[PER], uint32_t k, uint32_t* hist /*[256]*/, uint32_t* sel /*[2]*/) {
  if (threadIdx.x == 0) { sel[0] = 0; sel[1] = k; }
  uint32_t mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t prefix = sel[0];
#pragma unroll
    for (int e = 0; e < PER; ++e)
      if ((key[e] & mask) == prefix) atomicAdd(&hist[(key[e] >> shift) & 255], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {  // one wave: suffix sums over the 256 bins (4 bins per lane), find the bin holding the k-th
      const int l = threadIdx.x;
      const uint32_t h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
      uint32_t incl = h0 + h1 + h2 + h3;
      const uint32_t own = incl;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_down(incl, o, 64);
        if (l + o < 64) incl += v;
      }
      const uint32_t above = incl - own;  // keys in bins of higher lanes
      const uint32_t kk = sel[1];
      const uint32_t c3 = above + h3, c2 = c3 + h2, c1 = c2 + h1, c0 = c1 + h0;
      if (above < kk && kk <= c0) {
        int bin; uint32_t before;
        if (kk <= c3) { bin = 4 * l + 3; before = above; }
        else if (kk <= c2) { bin = 4 * l + 2; before = c3; }
        else if (kk <= c1) { bin = 4 * l + 1; before = c2; }
        else { bin = 4 * l; before = c1; }
        sel[0] = prefix | ((uint32_t)bin << shift);
        sel[1] = kk - before;
      }
    }
    mask |= 0xFFu << shift;
    __syncthreads();
  }
  return sel[0];
}

// Selection over one dense score row per query (<= 8192 columns).
//  BOOTSTRAP: only the k-th largest score matters (a lower bound of the final k-th best); thr = next_below(it).
//  otherwise (tiny corpora): radix-select the k-th score, gather the keys at or above it (k plus ties), sort that
//  handful by (score desc, id asc) and write the k best to the running list.
template <bool BOOTSTRAP>
__global__ __launch_bounds__(1024) void dense_select_kernel(SelectArgs a) {
  __shared__ uint64_t keys[kSelectCap];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sel[2];
  __shared__ int fill_s;
  const uint32_t q = blockIdx.x;
  if (q >= a.nq) return;
  const int n = (int)a.dense_cols;
  constexpr int PER = kSelectCap / 1024;
  uint32_t ordk[PER], ids[PER];
  const uint32_t sel_bits = a.seg.n ? a.seg.sel[q] : 0u;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int c = threadIdx.x + e * 1024;
    uint32_t o = 0, row = 0;
    if (c < n) {
      row = (a.tile_first + (uint32_t)(c >> 5) * a.tile_stride) * kTileRows + (c & 31);
      const float s = a.dense[(size_t)q * a.dense_ld + c];
      const bool counts = a.seg.n ? seg_row_counts(a.seg, row, sel_bits) : row < a.n_rows;
      if (counts) o = ord_f32(s);  // NaN -> 0 = absent
    }
    ordk[e] = o;
    ids[e] = row;
  }
  if (threadIdx.x == 0) fill_s = 0;
  const uint32_t kth = radix_select_kth<PER>(ordk, (uint32_t)a.k, hist, sel);  // ends with a barrier
  if (BOOTSTRAP) {
    if (threadIdx.x == 0 && kth != 0 && a.k <= n) {   // (sample rows may tie with the k-th: they pass in both modes)
      if (a.seg.n == 0) a.thr[q] = next_below(unord_f32(kth));
      else publish_thr(a, q, unord_f32(kth));
    }
    return;
  }
  // kth == 0: fewer than k valid rows -> keep every valid one
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    if (ordk[e] != 0 && ordk[e] >= kth) {
      const int pos = atomicAdd(&fill_s, 1);
      keys[pos] = ((uint64_t)ordk[e] << 32) | (uint64_t)(0xFFFFFFFFu - ids[e]);
    }
  }
  __syncthreads();
  const int fill = fill_s;
  const int np = pow2_ceil(fill < 2 ? 2 : fill);
  for (int i = fill + threadIdx.x; i < np; i += blockDim.x) keys[i] = 0;
  __syncthreads();
  bitonic_sort_desc(keys, np);
  const int cnt = fill < a.k ? fill : a.k;
  for (int i = threadIdx.x; i < a.k; i += blockDim.x) a.list[(size_t)q * a.list_ld + i] = i < cnt ? keys[i] : 0;
  if (threadIdx.x == 0) a.list_cnt[q] = cnt;
}

template <bool FINAL>
__global__ __launch_bounds__(1024) void compact_kernel(SelectArgs a, FinalizeArgs fin) {
  __shared__ uint64_t keys[kSelectCap];
  __shared__ int scan[16];
  const uint32_t q = blockIdx.x;
  if (q >= a.nq) return;
  const int tid = threadIdx.x;
  const int k = a.k, cap = a.cap;
  int fill = (int)a.list_cnt[q];
  for (int i = tid; i < fill; i += blockDim.x) keys[i] = a.list[(size_t)q * a.list_ld + i];
  auto sort_truncate = [&]() {
    const int np = pow2_ceil(fill < 2 ? 2 : fill);
    for (int i = fill + tid; i < np; i += blockDim.x) keys[i] = 0;
    __syncthreads();
    bitonic_sort_desc(keys, np);
    if (fill > k) fill = k;
  };
  // workgroup-wide exclusive scan of one value per thread (wave shuffles + one LDS hop); returns the total
  auto block_scan = [&](int c, int& offs) -> int {
    const int lane = tid & 63, w = tid >> 6;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) scan[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < 16; ++i) {
      const int v = scan[i];
      if (i < w) base += v;
      tot += v;
    }
    __syncthreads();
    offs = base + incl - c;
    return tot;
  };
  // fast path: every buffer of this query fits next to the running list (the common case: a few hundred survivors)
  bool done = false;
  if (a.nbuf <= 2048) {   // (one buffer per thread; two where a workgroup owns 8 buffers per query: the row-split wide-row kernel)
    int c0 = 0, c1 = 0;
    if ((uint32_t)tid < a.nbuf) {
      c0 = (int)a.cand_cnt[q * a.nbuf + tid];
      if (c0 > cap) c0 = cap;
    }
    if ((uint32_t)tid + 1024u < a.nbuf) {
      c1 = (int)a.cand_cnt[q * a.nbuf + tid + 1024];
      if (c1 > cap) c1 = cap;
    }
    int offs;
    const int tot = block_scan(c0 + c1, offs);
    if (fill + tot <= kSelectCap) {
      const uint64_t* src = a.cand + ((size_t)q * a.nbuf + tid) * cap;
      for (int e = 0; e < c0; ++e) keys[fill + offs + e] = src[e];
      src += (size_t)1024 * cap;
      for (int e = 0; e < c1; ++e) keys[fill + offs + c0 + e] = src[e];
      fill += tot;
      __syncthreads();
      done = true;
    }
  }
  int G = (kSelectCap - kMaxK) / cap;  // buffers appended per round; a round always fits after a truncation
  if (G > 1024) G = 1024;
  for (uint32_t g0 = 0; !done && g0 < a.nbuf; g0 += G) {
    const uint32_t b = g0 + tid;
    int c = 0;
    if (tid < G && b < a.nbuf) {
      c = (int)a.cand_cnt[q * a.nbuf + b];
      if (c > cap) c = cap;
    }
    int offs;
    const int tot = block_scan(c, offs);
    if (tot == 0) continue;
    if (fill + tot > kSelectCap) sort_truncate();  // uniform: fill and tot are workgroup-uniform
    if (c > 0) {
      const uint64_t* src = a.cand + ((size_t)q * a.nbuf + b) * cap;
      for (int e = 0; e < c; ++e) keys[fill + offs + e] = src[e];
    }
    fill += tot;
    __syncthreads();
  }
  sort_truncate();
  const uint64_t* sorted = keys;
  if (FINAL) {
    const int cnt = (fin.mask && !fin.mask[(size_t)q * fin.mask_stride]) ? 0 : fill;
    for (int i = tid; i < k; i += blockDim.x) {
      float sc = -__builtin_inff();
      int64_t id = -1;
      if (i < cnt) {
        sc = key_score(sorted[i]);
        id = a.seg.n ? seg_result_id(a.seg, key_id(sorted[i])) : (int64_t)key_id(sorted[i]) + fin.id_offset;
      }
      fin.D[(size_t)q * k + i] = sc;
      fin.I[(size_t)q * k + i] = id;
    }
    return;
  }
  for (int i = tid; i < k; i += blockDim.x) a.list[(size_t)q * a.list_ld + i] = i < fill ? sorted[i] : 0;
  if (tid == 0) {
    a.list_cnt[q] = fill;
    // strict threshold: every later row has a larger id than the k listed rows, so a tie loses
    if (fill == k) publish_thr(a, q, key_score(sorted[k - 1]));
  }
}

// L2 = the list holds q.x - |x|^2/2; the reported distance is |q|^2 - 2 * that (squared L2, as faiss.IndexFlatL2), padding +inf.
template <typename T, bool L2>
__global__ __launch_bounds__(256) void finalize_kernel(SelectArgs a, float* D, int64_t* I, int64_t id_offset, const uint8_t* mask,
                                                       int64_t mask_stride, const T* xq, int dim) {
  __shared__ float red[4];
  const uint32_t q = blockIdx.x;
  if (q >= a.nq) return;
  float qn = 0.f;
  if (L2) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) {
      const float v = (float)xq[(size_t)q * dim + i];
      acc += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    qn = (red[0] + red[1]) + (red[2] + red[3]);
  }
  const int cnt = (mask && !mask[(size_t)q * mask_stride]) ? 0 : (int)a.list_cnt[q];
  for (int i = threadIdx.x; i < a.k; i += blockDim.x) {
    float s = L2 ? __builtin_inff() : -__builtin_inff();
    int64_t id = -1;
    if (i < cnt) {
      const uint64_t key = a.list[(size_t)q * a.list_ld + i];
      s = key_score(key);
      if (L2) s = qn - 2.f * s;
      id = a.seg.n ? seg_result_id(a.seg, key_id(key)) : (int64_t)key_id(key) + id_offset;
    }
    D[(size_t)q * a.k + i] = s;
    I[(size_t)q * a.k + i] = id;
  }
}

// ---- cross-source merge (K4) ---------------------------------------------------------------
// (ord', id) pairs with 64-bit ids: ord' = ord(score) for descending, ~ord(score) for ascending,
// 0 for padding / NaN.  better(a,b) = a.s > b.s || (a.s == b.s && a.id < b.id).
// Scores compare as IEEE f32: -0.0 == +0.0 (what numpy's argsort of rerank.py:5,30 does), so a zero is keyed as +0.0 and its
// sign rides in bit 63 of the id slot (valid ids are >= 0), masked in every comparison and restored on output.
constexpr uint64_t kNegZeroBit = 1ull << 63;
// Candidate c of query q sits in list c / k_in (lists = n_outer x n_inner blocks of [nq][k_in], see MergeSrc), entry c % k_in:
// the plain [nq][m] form is one list of k_in = m; the gathered exchange buffer is read where the collective left it.
__global__ __launch_bounds__(1024) void merge_topk_kernel(const MergeSrc src, int m, int k, int descending, float* Dout, int64_t* Iout) {
  __shared__ uint32_t ss[kSelectCap];
  __shared__ int64_t ids[kSelectCap];
  const uint32_t q = blockIdx.x;
  const int np = pow2_ceil(m < 2 ? 2 : m);
  for (int c = threadIdx.x; c < np; c += blockDim.x) {
    uint32_t s = 0;
    int64_t id = INT64_MAX;
    if (c < m) {
      const int list = c / src.k_in, j = c - list * src.k_in;
      const int o = list / src.n_inner, in = list - o * src.n_inner;
      const size_t at = (size_t)in * src.inner + (size_t)q * src.k_in + j;
      const float v = src.D[(size_t)o * src.outer_D + at];
      const int64_t i = src.I[(size_t)o * src.outer_I + at];
      if (i >= 0 && v == v) {
        const uint32_t o = ord_f32(v == 0.f ? 0.f : v);
        s = descending ? o : ~o;
        id = (v == 0.f && (__float_as_uint(v) >> 31)) ? (int64_t)((uint64_t)i | kNegZeroBit) : i;
      }
    }
    ss[c] = s;
    ids[c] = id;
  }
  __syncthreads();
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int k2 = 2; k2 <= np; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np; i += nt) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint32_t sx = ss[i], sy = ss[ixj];
          const int64_t rx = ids[i], ry = ids[ixj];
          const int64_t ix = rx & INT64_MAX, iy = ry & INT64_MAX;
          const bool x_worse = sx < sy || (sx == sy && ix > iy);
          const bool desc = (i & k2) == 0;
          if (x_worse == desc && !(sx == sy && ix == iy)) {
            ss[i] = sy; ss[ixj] = sx; ids[i] = ry; ids[ixj] = rx;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    float v = descending ? -__builtin_inff() : __builtin_inff();
    int64_t id = -1;
    if (i < np && ss[i] != 0) {
      v = unord_f32(descending ? ss[i] : ~ss[i]);
      id = ids[i];
      if (id < 0) { v = -0.f; id &= INT64_MAX; }
    }
    Dout[(size_t)q * k + i] = v;
    Iout[(size_t)q * k + i] = id;
  }
}

hipError_t launch_prep(const SelectArgs& a, const void* xq, void* xqs, int dim, int blocks_per_wave, hipStream_t st,
                       const SegHost* segs, const uint8_t* route_mask, int64_t mask_stride) {
  const bool swz = xqs && blocks_per_wave > 0 && dim % 32 == 0;
  // wide rows, at most 16 queries: the few-query kernels read block 0 only (the reference's call shape is one query)
  const int n_blocks = (dim > 2 * kMaxResidentDim && a.nq <= 16) ? 1 : 4 * blocks_per_wave;
  const int blocks = swz ? (n_blocks * (dim / 32) * 64 + 255) / 256 : 1;
  SegHost none;
  memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(prep_kernel, dim3(blocks), dim3(256), 0, st, a, (const uint4*)xq, swz ? (uint4*)xqs : nullptr, dim, n_blocks,
                     segs ? *segs : none, route_mask, mask_stride);
  return hipGetLastError();
}
hipError_t launch_dense_select(const SelectArgs& a, bool bootstrap, hipStream_t st) {
  if (bootstrap) hipLaunchKernelGGL(dense_select_kernel<true>, dim3(kQueriesPerBlock), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL(dense_select_kernel<false>, dim3(kQueriesPerBlock), dim3(1024), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_compact(const SelectArgs& a, const FinalizeArgs* fin, hipStream_t st) {
  if (fin) hipLaunchKernelGGL(compact_kernel<true>, dim3(kQueriesPerBlock), dim3(1024), 0, st, a, *fin);
  else hipLaunchKernelGGL(compact_kernel<false>, dim3(kQueriesPerBlock), dim3(1024), 0, st, a, FinalizeArgs{});
  return hipGetLastError();
}
hipError_t launch_finalize(const SelectArgs& a, float* D, int64_t* I, int64_t id_offset, const uint8_t* mask, int64_t mask_stride,
                           const void* xq_l2, int dtype, int dim, hipStream_t st) {
  if (!xq_l2)
    hipLaunchKernelGGL((finalize_kernel<_Float16, false>), dim3(kQueriesPerBlock), dim3(256), 0, st, a, D, I, id_offset, mask, mask_stride,
                       (const _Float16*)nullptr, dim);
  else if (dtype == RR_DTYPE_F16)
    hipLaunchKernelGGL((finalize_kernel<_Float16, true>), dim3(kQueriesPerBlock), dim3(256), 0, st, a, D, I, id_offset, mask, mask_stride,
                       (const _Float16*)xq_l2, dim);
  else
    hipLaunchKernelGGL((finalize_kernel<__bf16, true>), dim3(kQueriesPerBlock), dim3(256), 0, st, a, D, I, id_offset, mask, mask_stride,
                       (const __bf16*)xq_l2, dim);
  return hipGetLastError();
}
hipError_t launch_merge_topk(const MergeSrc& src, int nq, int m, int k, int descending, float* Dout, int64_t* Iout, hipStream_t st) {
  hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(1024), 0, st, src, m, k, descending, Dout, Iout);
  return hipGetLastError();
}

}  // namespace rr
