// K1: brute-force inner-product scan of an HBM-resident corpus against 256 resident queries,
// fused with the top-k candidate filter.  Replaces the arithmetic inside
// `index.search(query_embed, k)` (reference ragroute/data_source.py:158,186,203).
//
// Design (gfx950 / MI355X, one persistent 256-thread workgroup per CU, one wave per SIMD):
//  * The 256 queries never leave the register file: each wave keeps 64 queries x D halves as
//    MFMA B-operand fragments (D=768 -> 384 of its 512 VGPR/AGPRs).  4 waves x 64 = 256.
//  * The corpus is streamed exactly once, HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) in
//    32-row tiles through a 3-slot ring (2 tiles in flight per CU); every 1 KiB DMA piece is
//    8 rows x 128 contiguous bytes (whole cache lines), XOR-swizzled on the SOURCE side so the
//    later ds_read_b128 of MFMA A-fragments is bank-conflict-free.
//  * Per tile: 48 x { ds_read_b128 corpus fragment ; 4 x v_mfma_f32_16x16x32 } with the corpus as A and the queries
//    as B, so every lane ends up owning ONE query (column) and a few corpus rows per result block: the top-k filter
//    is a per-lane compare against that query's threshold, no cross-lane work.  No inter-workgroup reuse exists
//    (every byte is read once), so no XCD-aware block remap.
//  * Scores strictly above the threshold are appended (as 64-bit order keys) to a private
//    per-(workgroup, query, lane-quarter) buffer; if a buffer fills, the wave compacts it exactly
//    to its k best and raises that lane's threshold.  Nothing is ever dropped that could be in
//    the final top-k (see DESIGN.md "exactness").
//  * DENSE=true writes every score instead (bootstrap sample and tiny corpora).
//
// Kernels in this file (launch_flat_scan picks one):
//   flat_scan16_kernel      D <= 768: 16x16x32 MFMA, hand-pipelined asm loop, nt LDS-DMA, optional L2 metric; also the int8
//                           screening copy (I8Pair, v_mfma_i32_16x16x64_i8)
//   flat_scan16h_kernel     768 < D <= 1536, up to 128 queries: 32 resident queries per wave, half-tile LDS ring
//   flat_scan_wide_kernel   D > 1536, and more than 128 queries (or the L2 metric) at 768 < D <= 1536: accumulators resident
//                           in AGPRs, corpus by LDS-DMA, queries streamed from L2
// The measured alternatives (32x32x16 shape, compiler-scheduled loop, two waves per SIMD, compiler-scheduled wide-row forms,
// stamped / ablation builds) live in flat_scan_dev.hip and are compiled only with RR_DEV_VARIANTS=1 (see _build.py).
#include "flat_scan_common.h"

namespace rr {

// L2 = true ranks by  q.x - |x|^2/2  (descending == ascending squared L2 distance): the per-row |x|^2/2 of a tile is
// one more 256-byte LDS-DMA piece (issued by wave 0, ahead of the tile-after-next's pieces so the in-order vmcnt wait of
// the tile covers it) and the epilogue subtracts it before the filter.
#ifndef RR_SCAN_STAGGER
#define RR_SCAN_STAGGER 0
#endif
// SEG = true: the launch is a chunk of a segmented search (a.ranges: one run of tiles per segment, a TileCursor per stream).  Plain
// searches run the SEG = false instantiation, whose tile arithmetic and epilogue carry nothing of that (round 4: the same-device
// A/B against round 2's kernel priced the run-time form at 0.5 - 0.8 % of the headline's scan launch, profiles/r04/headline_ab.json).
template <typename T, int D, bool DENSE, bool NT = false, bool L2 = false, bool SEG = false>
__global__ __launch_bounds__(256, 1) void flat_scan16_kernel(const ScanArgs a) {
  typedef typename Mfma<T>::frag frag;
  constexpr int KS2 = D / 32;        // 32-wide k slices
  constexpr int KG = D / 64;         // 64-wide k groups (DMA pieces)
  constexpr int NF = 2 * KS2;        // corpus fragments per tile (2 row blocks per slice)
  constexpr int NQ = 4 * KS2;        // resident query fragments per wave
  constexpr int NAQ = NQ < 64 ? NQ : 64;  // ... of which live in AGPRs (all 256)
  constexpr int TILE_BYTES = kTileRows * D * 2;
  constexpr int NS = scan16_slots(D);  // LDS ring slots; NS - 1 tiles are in flight while one is computed
  constexpr int INFL = NS - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, g = lane >> 4;
  if (a.timeline && threadIdx.x == 0) a.timeline[blockIdx.x] = __builtin_amdgcn_s_memrealtime();  // diagnostics only

  // ---- LDS image (identical to flat_scan_kernel's).  A fragment (rb, s2): row rb*16 + col, chunk 4*(s2&1) + g of
  // k group s2>>1 -------------------------------------------------------------------------------------------------
  uint32_t roff[2][2];
  {
    const int rho = col & 7;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int p = 2 * rb + (col >> 3);
      const int f = ((rho >> 1) & 3) | ((p & 1) << 2);
#pragma unroll
      for (int par = 0; par < 2; ++par) roff[rb][par] = p * 1024 + rho * 128 + (((4 * par + g) ^ f) * 16);
    }
  }
  const int rho_w = lane >> 3, sig = lane & 7;
  const int f_w = ((rho_w >> 1) & 3) | ((wave & 1) << 2);
  const int c_w = sig ^ f_w;
  TileCursor dcur;   // the DMA stream's position in the launch's tile runs (segmented search; plain: tile_first + j * tile_stride)
  if (SEG) cursor_init(a, dcur);
  auto tile_src = [&](uint32_t j) -> const char* {
    if (j >= a.n_tiles) j = a.n_tiles - 1;
    const uint32_t tile = SEG ? cursor_tile(a, dcur, j) : a.tile_first + j * a.tile_stride;
    uint32_t row = tile * kTileRows + wave * 8 + rho_w;
    row = row < a.n_rows ? row : a.n_rows - 1;
    return (const char*)a.xb + (size_t)row * (D * 2) + c_w * 16;
  };
  auto issue_piece = [&](const char* gp, int slot, int kg) {
    char* l = smem + slot * TILE_BYTES + wave * 1024;
    if (NT)  // non-temporal: the corpus is read once per batch, keep it out of L2 / Infinity Cache
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + kg * 128),
                                       (__attribute__((address_space(3))) void*)(l + kg * 4096), 16, 0, 2);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + kg * 128),
                                       (__attribute__((address_space(3))) void*)(l + kg * 4096), 16, 0, 0);
  };

  // The batch is dealt to the four waves in blocks of 16 queries, round robin: slot qb of wave w holds block 4 qb + w, so 128
  // queries are 2 blocks on every SIMD instead of 4 on two of them (the step is bound by the busiest wave's MFMAs).
  // Real query blocks of this wave (wave-uniform): slots 0 .. nb-1
  const int nblk = ((int)a.nq + 15) >> 4;
  const int nb = __builtin_amdgcn_readfirstlane(nblk > wave ? (nblk - wave + 3) >> 2 : 0);

  // L2: |x|^2/2 of the 32 rows of tile ordinal jj -> LDS floats [NS*TILE_BYTES + slot*256 ...] (lanes 32..63 duplicate)
  auto issue_norms = [&](uint32_t jj, int slot_) {   // (L2 metric: never a segmented search, no cursor)
    if (jj >= a.n_tiles) jj = a.n_tiles - 1;
    uint32_t row = (a.tile_first + jj * a.tile_stride) * kTileRows + (lane & 31);
    row = row < a.n_rows ? row : a.n_rows - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.half_sqnorm + row),
                                     (__attribute__((address_space(3))) void*)(smem + NS * TILE_BYTES + slot_ * 256), 4, 0, 0);
  };
  const bool norm_wave = L2 && wave == 1;
  // Tile schedule: static round-robin, ordinals blockIdx, +grid, +2 grid, ...  (Ticketed dynamic tiles - XCDs run up to 7.5 %
  // apart under load - levelled the finish times without shortening the launch and were removed; see DESIGN.md.)
  const uint32_t stride = gridDim.x;
  const uint32_t n_tiles = a.n_tiles;
  uint32_t j = blockIdx.x;
  if (j < n_tiles) {
#pragma unroll
    for (int t = 0; t < INFL; ++t) {
      const char* gt = tile_src(j + t * stride);
      if (norm_wave) issue_norms(j + t * stride, t);
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) issue_piece(gt, t, kg);
    }
  }

  // (the first two tiles are already in flight: their HBM latency overlaps the query loads below)
  // ---- resident queries: B fragment (qb, s2): query (4 qb + wave)*16 + col, k = 32 s2 + 8 g .. +7 -------------
  // (read from the fragment-order copy the prep kernel made, a.xqs: every load is one contiguous KiB per wave)
  frag q[4][KS2];
  {
    const T* xqs = (const T*)a.xqs;
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      const T* p = xqs + ((size_t)(qb * 4 + wave) * KS2 * 64 + lane) * 8;
#pragma unroll
      for (int s2 = 0; s2 < KS2; ++s2) {
        if (qb * KS2 + s2 < NAQ) agpr_load_frag(q[qb][s2], p + 512 * s2);
        else q[qb][s2] = *(const frag*)(p + 512 * s2);
      }
    }
#pragma unroll
    for (int i = 0; i < NAQ; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+a"(q[i / KS2][i % KS2]));
  }

  LaneState4 st;
  lane_state_segments_init(a, st);
  const uint32_t nbuf = gridDim.x * 4;
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const uint32_t qi = (qb * 4 + wave) * 16 + col;
    st.thr[qb] = DENSE ? 0.f : a.thr[qi];
    st.cnt[qb] = 0;
    st.off[qb] = (qi * nbuf + blockIdx.x * 4 + g) * (uint32_t)a.cap;
  }

  // The initial thresholds are global loads: make them land here (the queue was drained for the query loads anyway).  Left to
  // hipcc, their first use inside the loop gets an s_waitcnt vmcnt(0) that then drains the DMA ring on EVERY tile.
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(st.thr[0]), "+v"(st.thr[1]), "+v"(st.thr[2]), "+v"(st.thr[3]));
  int slot = 0;
  uint32_t dbg_iter = 0;
  while (j < n_tiles) {
    // queue (oldest first): [norms t] DMA t  [norms t+1] DMA t+1 ... -> all but the pieces of the INFL-1 younger tiles
    // (KG each, +1 on the norm wave) are done
    if (norm_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((INFL - 1) * (KG + 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((INFL - 1) * KG) : "memory");
    if (a.timeline && threadIdx.x == 0) {  // diagnostics only: tile sequence, and the time the first tile became ready
      if (dbg_iter == 0) a.timeline[3 * gridDim.x + 64 * gridDim.x + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
      a.timeline[3 * gridDim.x + blockIdx.x * 64 + (dbg_iter & 63)] = j;
    }
    if (a.timeline) ++dbg_iter;
    __builtin_amdgcn_s_barrier();
#if RR_SCAN_STAGGER
    // the four waves leave the barrier together and share one address unit: start wave w one MFMA time (16 cycles) x w late so
    // that their DMA instructions, which sit at the same places of the same code, do not arrive there together
    if (wave & 1) asm volatile("s_nop 15" ::: "memory");
    if (wave & 2) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
    int nslot = slot + INFL;
    if (nslot >= NS) nslot -= NS;
    uint32_t j2 = j + INFL * stride;
    const char* gn = tile_src(j2);
    if (norm_wave) issue_norms(j2, nslot);
    auto compute = [&](auto tag) {
      constexpr int NQB = decltype(tag)::value;
      f32x4 acc[2][4];
      constexpr int NB = NF < 8 ? NF : 8;
      frag c[NB];
      uint32_t ab[2][2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int par = 0; par < 2; ++par) ab[rb][par] = (uint32_t)(slot * TILE_BYTES) + roff[rb][par];
      // fragment index f = 2*s2 + rb
#pragma unroll
      for (int f = 0; f < NB; ++f) lds_read_frag(c[f], ab[f & 1][(f >> 1) & 1], (f >> 2) * 4096);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int s2 = f >> 1, rb = f & 1;
        if (NF - f >= NB) lgkm_wait<NB - 1>();
        else if (NF - f == 7) lgkm_wait<6>();
        else if (NF - f == 6) lgkm_wait<5>();
        else if (NF - f == 5) lgkm_wait<4>();
        else if (NF - f == 4) lgkm_wait<3>();
        else if (NF - f == 3) lgkm_wait<2>();
        else if (NF - f == 2) lgkm_wait<1>();
        else lgkm_wait<0>();
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
          const bool in_a = qb * KS2 + s2 < NAQ;
          if (s2 == 0) {
            if (in_a) Mfma16Asm<T>::first_a(acc[rb][qb], c[f % NB], q[qb][0]);
            else Mfma16Asm<T>::first_v(acc[rb][qb], c[f % NB], q[qb][0]);
          } else {
            if (in_a) Mfma16Asm<T>::acc_a(acc[rb][qb], c[f % NB], q[qb][s2]);
            else Mfma16Asm<T>::acc_v(acc[rb][qb], c[f % NB], q[qb][s2]);
          }
        }
        if (f + NB < NF) lds_read_frag(c[f % NB], ab[(f + NB) & 1][((f + NB) >> 1) & 1], ((f + NB) >> 2) * 4096);
        // one DMA piece every 4 fragments; other placements (before the ds_read, pairs every 8, all up front) measured
        // equal, equal and 5 % slower
        if ((f & 3) == 1) issue_piece(gn, nslot, f >> 2);
      }
      if (NQB == 4)
        asm volatile("s_nop 15\n\ts_nop 7"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                       "+v"(acc[1][2]), "+v"(acc[1][3]));
      else if (NQB == 3)
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]));
      else if (NQB == 2)
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
      else
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][0]));
      if (std::is_same<T, I8Pair>::value) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[rb][qb][i] = (float)__float_as_int(acc[rb][qb][i]);
      }
      if (L2) {
        const f32x4 h0 = lds_load_f32x4((uint32_t)(NS * TILE_BYTES + slot * 256 + (4 * g) * 4));
        const f32x4 h1 = lds_load_f32x4((uint32_t)(NS * TILE_BYTES + slot * 256 + (16 + 4 * g) * 4));
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
          acc[0][qb] -= h0;
          acc[1][qb] -= h1;
        }
      }
      tile_epilogue16<DENSE, NQB, 64, false, 4, true, SEG>(a, st, acc, j, lane, wave);
    };
    if (nb >= 4) {
      compute(std::integral_constant<int, 4>{});
    } else if (nb == 3) {
      compute(std::integral_constant<int, 3>{});
    } else if (nb == 2) {
      compute(std::integral_constant<int, 2>{});
    } else if (nb == 1) {
      compute(std::integral_constant<int, 1>{});
    } else {
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) issue_piece(gn, nslot, kg);
    }
    slot = slot + 1;
    if (slot >= NS) slot = 0;
    j += stride;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
  if (!DENSE) {
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) a.cand_cnt[((qb * 4 + wave) * 16 + col) * nbuf + blockIdx.x * 4 + g] = st.cnt[qb];
  }
  if (a.timeline && threadIdx.x == 0) {
    a.timeline[gridDim.x + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    a.timeline[2 * gridDim.x + blockIdx.x] = (uint64_t)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  }
}

// ---- 768 < D <= 1536 (FeB4RAG's 1024-wide encoders): resident queries, half-tile ring --------------------------
// 32 queries per wave stay resident (2 blocks of 16: D/4 registers), so one launch serves 128 queries and a 256-query
// block takes two passes over the corpus (each pass runs near the HBM rate because it carries half the MFMA work).
// The LDS ring works on 16-row half tiles (16 x D x 2 bytes: 32 KB at D = 1024 -> 4 slots, 48 KB at 1536 -> 3 slots).
template <typename T, int D, bool DENSE, bool NT>
__global__ __launch_bounds__(256, 1) void flat_scan16h_kernel(const ScanArgs a) {
  typedef typename Mfma<T>::frag frag;
  constexpr int KS2 = D / 32, KG = D / 64;
  constexpr int UNIT_BYTES = 16 * D * 2;
  constexpr int NSLOT = (160 * 1024) / UNIT_BYTES >= 4 ? 4 : 3;
  constexpr int AHEAD = NSLOT - 1;            // units in flight ahead of the one being multiplied
  constexpr int PIECES = KG / 2;              // DMA pieces per wave and unit: 2 row groups x KG k groups over 4 waves
  constexpr int NQ = 2 * KS2;
  constexpr int NAQ = NQ < 62 ? NQ : 62;
  static_assert(KG % 2 == 0, "D must be a multiple of 128");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, g = lane >> 4;

  // (resident queries come from the fragment-order copy of the prep kernel, a.xqs: [wave][2 blocks][k slice][lane][8])
  frag q[2][KS2];
  {
    const T* xqs = (const T*)a.xqs;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const T* p = xqs + ((size_t)(wave * 2 + qb) * KS2 * 64 + lane) * 8;
#pragma unroll
      for (int s2 = 0; s2 < KS2; ++s2) {
        if (qb * KS2 + s2 < NAQ) agpr_load_frag(q[qb][s2], p + 512 * s2);
        else q[qb][s2] = *(const frag*)(p + 512 * s2);
      }
    }
#pragma unroll
    for (int i = 0; i < NAQ; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+a"(q[i / KS2][i % KS2]));
  }
  float thr[2];
  uint32_t cnt[2], off[2];
  TileCursor dcur, ecur;                                               // segmented search: DMA-stream and epilogue cursors (see TileCursor)
  cursor_init(a, dcur);
  cursor_init(a, ecur);
  uint32_t row_limit = a.n_rows;
  const uint32_t nbuf = gridDim.x * 4;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const uint32_t qi = wave * 32 + qb * 16 + col;
    thr[qb] = DENSE ? 0.f : a.thr[qi];
    cnt[qb] = 0;
    off[qb] = (qi * nbuf + blockIdx.x * 4 + g) * (uint32_t)a.cap;
  }
  // A fragment of slice s2 in a 16-row unit: row col (row group p = col>>3), chunk 4*(s2&1) + g of k group s2>>1;
  // unit image: piece (kg, p) at (kg*2 + p) * 1024
  uint32_t roff[2];
  {
    const int p = col >> 3, rho = col & 7;
    const int f = ((rho >> 1) & 3) | (p << 2);
#pragma unroll
    for (int par = 0; par < 2; ++par) roff[par] = p * 1024 + rho * 128 + (((4 * par + g) ^ f) * 16);
  }
  // DMA: wave w fills row group p = w&1 of the k groups with parity w>>1
  const int pw = wave & 1, kpar = wave >> 1;
  const int rho_w = lane >> 3, sig = lane & 7;
  const int f_w = ((rho_w >> 1) & 3) | (pw << 2);
  const int c_w = sig ^ f_w;
  // unit ordinal u = 2*i + half, i-th tile of this workgroup (tile ordinal j = blockIdx + i*gridDim)
  const uint32_t my_tiles = a.n_tiles > blockIdx.x ? (a.n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const uint32_t n_units = 2 * my_tiles;
  auto unit_src = [&](uint32_t u) -> const char* {
    if (u >= n_units) u = n_units - 1;
    const uint32_t j = blockIdx.x + (u >> 1) * gridDim.x;
    const uint32_t tile = cursor_tile(a, dcur, j);
    uint32_t row = tile * kTileRows + (u & 1) * 16 + pw * 8 + rho_w;
    row = row < a.n_rows ? row : a.n_rows - 1;
    return (const char*)a.xb + (size_t)row * (D * 2) + c_w * 16 + kpar * 128;
  };
  auto issue_piece = [&](const char* gp, int slot, int i) {  // i-th piece: k group kpar + 2 i
    char* l = smem + slot * UNIT_BYTES + pw * 1024 + kpar * 2048;
    if (NT)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + i * 256),
                                       (__attribute__((address_space(3))) void*)(l + i * 4096), 16, 0, 2);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + i * 256),
                                       (__attribute__((address_space(3))) void*)(l + i * 4096), 16, 0, 0);
  };
  const int nb = __builtin_amdgcn_readfirstlane(
      (int)a.nq <= wave * 32 ? 0 : ((int)a.nq - wave * 32 >= 32 ? 2 : ((int)a.nq - wave * 32 + 15) / 16));

  if (n_units > 0) {
#pragma unroll
    for (int h = 0; h < AHEAD; ++h) {
      const char* g0 = unit_src(h);
#pragma unroll
      for (int i = 0; i < PIECES; ++i) issue_piece(g0, h, i);
    }
  }
  int slot = 0;
  for (uint32_t u = 0; u < n_units; ++u) {
    // unit u landed: all but the (AHEAD-1)*PIECES youngest DMA ops are done
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PIECES) : "memory");
    __builtin_amdgcn_s_barrier();
    int nslot = slot + AHEAD;
    if (nslot >= NSLOT) nslot -= NSLOT;
    const char* gn = unit_src(u + AHEAD);
    const uint32_t j = blockIdx.x + (u >> 1) * gridDim.x;
    if (nb > 0) {
      f32x4 acc[2];
      constexpr int NB = KS2 < 8 ? KS2 : 8;
      frag c[NB];
      uint32_t ab[2];
      ab[0] = (uint32_t)(slot * UNIT_BYTES) + roff[0];
      ab[1] = (uint32_t)(slot * UNIT_BYTES) + roff[1];
#pragma unroll
      for (int f = 0; f < NB; ++f) lds_read_frag(c[f], ab[f & 1], (f >> 1) * 2048);
#pragma unroll
      for (int f = 0; f < KS2; ++f) {
        if (KS2 - f >= NB) lgkm_wait<NB - 1>();
        else if (KS2 - f == 7) lgkm_wait<6>();
        else if (KS2 - f == 6) lgkm_wait<5>();
        else if (KS2 - f == 5) lgkm_wait<4>();
        else if (KS2 - f == 4) lgkm_wait<3>();
        else if (KS2 - f == 3) lgkm_wait<2>();
        else if (KS2 - f == 2) lgkm_wait<1>();
        else lgkm_wait<0>();
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const bool in_a = qb * KS2 + f < NAQ;
          if (f == 0) {
            if (in_a) Mfma16Asm<T>::first_a(acc[qb], c[0], q[qb][0]);
            else Mfma16Asm<T>::first_v(acc[qb], c[0], q[qb][0]);
          } else {
            if (in_a) Mfma16Asm<T>::acc_a(acc[qb], c[f % NB], q[qb][f]);
            else Mfma16Asm<T>::acc_v(acc[qb], c[f % NB], q[qb][f]);
          }
        }
        if (f + NB < KS2) lds_read_frag(c[f % NB], ab[(f + NB) & 1], ((f + NB) >> 1) * 2048);
        if ((f & 3) == 1 && (f >> 2) < PIECES) issue_piece(gn, nslot, f >> 2);
      }
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
      if (DENSE) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          *(f32x4*)(a.dense + (size_t)(wave * 32 + qb * 16 + col) * a.dense_ld + j * kTileRows + (u & 1) * 16 + 4 * g) = acc[qb];
      } else {
        if (j >= ecur.j_end) {   // segmented search: the wave enters another segment's slice (see tile_epilogue16)
          const RangeEntry e = cursor_advance(a, ecur, j);
          row_limit = e.row_limit;
          const uint32_t qi[4] = {(uint32_t)(wave * 32 + col), (uint32_t)(wave * 32 + 16 + col), 0u, 0u};
          float t4[4];
          load_thresholds<2>(a, e.seg, qi, t4);
          thr[0] = t4[0];
          thr[1] = t4[1];
        }
        const uint32_t row0 = (a.tile_first + j * a.tile_stride + (uint32_t)ecur.delta) * kTileRows + (u & 1) * 16 + 4 * g;
        const float m0 = max4v(acc[0]), m1 = max4v(acc[1]);
        if (__builtin_amdgcn_ballot_w64(m0 > thr[0] || m1 > thr[1])) {
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t id = row0 + i;
              if (acc[qb][i] > thr[qb] && id < row_limit) {
                a.cand[(size_t)off[qb] + cnt[qb]] = make_key(acc[qb][i], id);
                ++cnt[qb];
              }
            }
          }
          const uint32_t lim = (uint32_t)a.cap - 16u;
          if (__builtin_amdgcn_ballot_w64(cnt[0] > lim || cnt[1] > lim)) {
            uint64_t* scratch = a.scratch + (size_t)(blockIdx.x * 4 + wave) * a.cap;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
              uint64_t mask = __builtin_amdgcn_ballot_w64(cnt[qb] > lim);
              while (mask) {
                const int L = __builtin_ctzll(mask);
                mask &= mask - 1;
                const uint32_t o = __shfl(off[qb], L, 64);
                const int cn = (int)__shfl(cnt[qb], L, 64);
                const uint64_t kth = wave_compact(a.cand + (size_t)__builtin_amdgcn_readfirstlane(o), scratch,
                                                  __builtin_amdgcn_readfirstlane(cn), a.k, lane);
                if (lane == L) { cnt[qb] = a.k; thr[qb] = a.ties_pass ? next_below(key_score(kth)) : key_score(kth); }
              }
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) issue_piece(gn, nslot, i);
    }
    slot = slot + 1;
    if (slot >= NSLOT) slot = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!DENSE) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) a.cand_cnt[(wave * 32 + qb * 16 + col) * nbuf + blockIdx.x * 4 + g] = cnt[qb];
  }
}

template <typename T, int D>
static hipError_t launch_scan16h(const ScanArgs& a, bool dense, int grid, hipStream_t st) {
  constexpr int unit = 16 * D * 2;
  constexpr int nslot = (160 * 1024) / unit >= 4 ? 4 : 3;
  const size_t lds = (size_t)nslot * unit;
  const bool nt = !dense && (size_t)a.n_rows * D * 2 > (256ull << 20);
  hipError_t e;
#define RR_LAUNCH_H(DENSE_, NT_)                                                                                      \
  e = hipFuncSetAttribute((const void*)flat_scan16h_kernel<T, D, DENSE_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
  if (e != hipSuccess) return e;                                                                                      \
  hipLaunchKernelGGL((flat_scan16h_kernel<T, D, DENSE_, NT_>), dim3(grid), dim3(256), lds, st, a);
  if (dense) { RR_LAUNCH_H(true, false) }
  else if (nt) { RR_LAUNCH_H(false, true) }
  else { RR_LAUNCH_H(false, false) }
#undef RR_LAUNCH_H
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_scan_half_resident(const ScanArgs& a, int D, bool dense, int grid, hipStream_t st) {
  switch (D) {
    case 896: return launch_scan16h<T, 896>(a, dense, grid, st);
    case 1024: return launch_scan16h<T, 1024>(a, dense, grid, st);
    case 1280: return launch_scan16h<T, 1280>(a, dense, grid, st);
    case 1536: return launch_scan16h<T, 1536>(a, dense, grid, st);
    default: return hipErrorInvalidValue;
  }
}

int g_wide_min_queries = 129;  // 768 < D <= 1536: batches of at least this many queries take the wide-row kernel in one pass (RR_WIDE_MIN_QUERIES)
int g_scan_variant = 16;       // 16 = default; 15 = the same kernel without non-temporal DMA; other values: development kernels (RR_DEV_VARIANTS builds)
int g_generic_tall = 16;       // 16 = flat_scan_wide_kernel; 8 / 4 / 0: compiler-scheduled wide-row forms (RR_DEV_VARIANTS builds)

static void read_variant_env() {
  static const bool env_read = [] {
    if (const char* v = tuning_env("RR_SCAN_VARIANT")) g_scan_variant = atoi(v);
    if (const char* v = tuning_env("RR_GENERIC_TALL")) g_generic_tall = atoi(v);
    if (const char* v = tuning_env("RR_WIDE_MIN_QUERIES")) g_wide_min_queries = atoi(v);
#ifndef RR_DEV_VARIANTS
    if (g_scan_variant != 15) g_scan_variant = 16;  // the development kernels are not in this build
    g_generic_tall = 16;
#endif
    return true;
  }();
  (void)env_read;
}

static bool half_resident_dim(int D) { return D == 896 || D == 1024 || D == 1280 || D == 1536; }
// queries one scan launch serves: 256, or 128 where only 32 queries per wave stay resident (768 < D <= 1536) and the batch is small
int scan_queries_per_launch(int D, int nq) {
  read_variant_env();
  if (g_scan_variant == 3) return 256;
  return (half_resident_dim(D) && nq < g_wide_min_queries) ? 128 : 256;  // (an L2 search at these dims still works in 128-query blocks: each goes to the wide-row kernel)
}

// 16-query blocks each wave keeps resident in the kernel that will serve (D, nq, metric): 4 (d <= 768), 2 (768 < d <= 1536 and a
// small inner-product batch), 0 = the wide-row kernels, which stream the row-major queries.  The prep kernel lays the queries out
// accordingly (fragment order).
int scan_query_blocks_per_wave(int D, int nq, bool l2) {
  read_variant_env();
  if (D <= kMaxResidentDim) return 4;
  if (half_resident_dim(D) && nq < g_wide_min_queries && !l2) return 2;
  return RR_WIDE_QFRAG ? 4 : 0;   // wide-row kernels: the 16 blocks of a 256-query pass, same [block][k slice][lane][8] order
}

// candidate buffers per (workgroup, query) of the kernel that will serve this block: 4 lane quarters; the row-split wide-row
// kernel 8 (two waves share each query)
int scan_bufs_per_wg(int D, int nq, bool l2, int k) {
  read_variant_env();
#ifdef RR_DEV_VARIANTS
  if (!l2 && dev_scan_bufs_per_wg(D, g_scan_variant, g_generic_tall) != 4) return dev_scan_bufs_per_wg(D, g_scan_variant, g_generic_tall);
#endif
  if (half_resident_dim(D) && nq < g_wide_min_queries && !l2) return 4;
  return scan_wide_rowsplit(D, nq, l2, k) ? 8 : 4;
}

template <typename T, int D, bool DENSE, bool NT = false, bool L2 = false, bool SEG = false>
static hipError_t launch_scan16(const ScanArgs& a, int grid, hipStream_t st) {
  const size_t lds = scan16_slots(D) * ((size_t)kTileRows * D * 2 + 256);  // ring + L2 norm slots
  hipError_t e = hipFuncSetAttribute((const void*)flat_scan16_kernel<T, D, DENSE, NT, L2, SEG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((flat_scan16_kernel<T, D, DENSE, NT, L2, SEG>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <typename T, int D>
static hipError_t launch_scan_t(const ScanArgs& a, bool dense, int grid, hipStream_t st) {
  const bool nt = (size_t)a.n_rows * D * 2 > kNtThresholdBytes && g_scan_variant != 15;
  if (a.ranges) {       // chunk launch of a segmented search (inner product, filter launches only)
    if (a.half_sqnorm || dense) return hipErrorNotSupported;
    return nt ? launch_scan16<T, D, false, true, false, true>(a, grid, st) : launch_scan16<T, D, false, false, false, true>(a, grid, st);
  }
  if (a.half_sqnorm) {  // L2 metric
    if (dense) return launch_scan16<T, D, true, false, true>(a, grid, st);
    return nt ? launch_scan16<T, D, false, true, true>(a, grid, st) : launch_scan16<T, D, false, false, true>(a, grid, st);
  }
  if (dense) return launch_scan16<T, D, true, false>(a, grid, st);
  return nt ? launch_scan16<T, D, false, true>(a, grid, st) : launch_scan16<T, D, false, false>(a, grid, st);
}

template <typename T>
static hipError_t launch_scan_d(const ScanArgs& a, int D, bool dense, int grid, hipStream_t st) {
  switch (D) {
    case 128: return launch_scan_t<T, 128>(a, dense, grid, st);
    case 256: return launch_scan_t<T, 256>(a, dense, grid, st);
    case 384: return launch_scan_t<T, 384>(a, dense, grid, st);
    case 512: return launch_scan_t<T, 512>(a, dense, grid, st);
    case 640: return launch_scan_t<T, 640>(a, dense, grid, st);
    case 768: return launch_scan_t<T, 768>(a, dense, grid, st);
    default: return hipErrorInvalidValue;
  }
}

// int8 screening scan: D = bytes per row / 2
static hipError_t launch_scan_i8(const ScanArgs& a, int D, bool dense, int grid, hipStream_t st) {
  const bool nt = (size_t)a.n_rows * D * 2 > kNtThresholdBytes;
#define RR_I8_CASE(D_)                                                            \
  case D_:                                                                        \
    if (dense) return launch_scan16<I8Pair, D_, true, false>(a, grid, st);        \
    return nt ? launch_scan16<I8Pair, D_, false, true>(a, grid, st) : launch_scan16<I8Pair, D_, false, false>(a, grid, st);
  switch (D) {
    RR_I8_CASE(128) RR_I8_CASE(256) RR_I8_CASE(384) RR_I8_CASE(512) RR_I8_CASE(640) RR_I8_CASE(768)
    default: return hipErrorInvalidValue;
  }
#undef RR_I8_CASE
}

hipError_t launch_flat_scan(const ScanArgs& a, int dtype, int D, bool dense, int grid, hipStream_t st) {
  read_variant_env();
  if (dtype == kDtypeI8) return a.half_sqnorm ? hipErrorNotSupported : launch_scan_i8(a, D, dense, grid, st);
  if (dtype != RR_DTYPE_F16 && dtype != RR_DTYPE_BF16) return hipErrorInvalidValue;
#ifdef RR_DEV_VARIANTS
  if (dev_scan_handles(a, D, g_scan_variant, g_generic_tall)) {
    if (a.ranges) return hipErrorNotSupported;   // the development kernels have no tile cursor: plain searches only
    return launch_dev_scan(a, dtype, D, dense, grid, st, g_scan_variant, g_generic_tall);
  }
#endif
  const bool f16 = dtype == RR_DTYPE_F16;
  if (half_resident_dim(D) && (int)a.nq < g_wide_min_queries && !a.half_sqnorm)  // 768 < D <= 1536, small batch: 32 resident queries per wave
    return f16 ? launch_scan_half_resident<_Float16>(a, D, dense, grid, st) : launch_scan_half_resident<__bf16>(a, D, dense, grid, st);
  if (D > kMaxResidentDim)  // wide rows (flat_scan_wide.hip); at 768 < D <= 1536 one pass of it beats two passes of the half-resident kernel
    return launch_scan_wide(a, dtype, D, dense, grid, st);
  return f16 ? launch_scan_d<_Float16>(a, D, dense, grid, st) : launch_scan_d<__bf16>(a, D, dense, grid, st);
}

// the kernel launch_flat_scan dispatches a filter launch of (D, nq queries in the block) to
const char* scan_kernel_name(int D, int nq, bool l2) {
  read_variant_env();
  if (half_resident_dim(D) && nq < g_wide_min_queries && !l2) return "flat_scan16h_kernel";
  if (D > kMaxResidentDim) return scan_wide_kernel_name(D, nq);
  return "flat_scan16_kernel";
}

int scan_padded_dim(int d) {
  static const int dims[] = {128, 256, 384, 512, 640, 768, 896, 1024, 1280, 1536};  // query-resident instantiations
  for (int v : dims)
    if (d <= v) return v;
  if (d <= kMaxDim) return (d + 127) / 128 * 128;  // wide-row kernel
  return -1;
}

}  // namespace rr
