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
template <typename T, int D, bool DENSE, bool NT = false, bool L2 = false>
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
  auto tile_src = [&](uint32_t j) -> const char* {
    if (j >= a.n_tiles) j = a.n_tiles - 1;
    const uint32_t tile = a.tile_first + j * a.tile_stride;
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

  // real query blocks of this wave (wave-uniform)
  const int nb = __builtin_amdgcn_readfirstlane(
      (int)a.nq <= wave * 64 ? 0 : ((int)a.nq - wave * 64 >= 64 ? 4 : ((int)a.nq - wave * 64 + 15) / 16));

  // L2: |x|^2/2 of the 32 rows of tile ordinal jj -> LDS floats [NS*TILE_BYTES + slot*256 ...] (lanes 32..63 duplicate)
  auto issue_norms = [&](uint32_t jj, int slot_) {
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
  // ---- resident queries: B fragment (qb, s2): query wave*64 + qb*16 + col, k = 32 s2 + 8 g .. +7 -------------
  frag q[4][KS2];
  {
    const T* xq = (const T*)a.xq;
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      const uint32_t qi = wave * 64 + qb * 16 + col;
      const T* p = xq + (size_t)(qi < a.nq ? qi : a.nq - 1) * D + 8 * g;
#pragma unroll
      for (int s2 = 0; s2 < KS2; ++s2) {
        if (qb * KS2 + s2 < NAQ) agpr_load_frag(q[qb][s2], p + 32 * s2);
        else q[qb][s2] = *(const frag*)(p + 32 * s2);
      }
    }
#pragma unroll
    for (int i = 0; i < NAQ; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+a"(q[i / KS2][i % KS2]));
  }

  LaneState4 st;
  const uint32_t nbuf = gridDim.x * 4;
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const uint32_t qi = wave * 64 + qb * 16 + col;
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
      tile_epilogue16<DENSE, NQB>(a, st, acc, j, lane, wave);
    };
    if (nb >= 2) {
      compute(std::integral_constant<int, 4>{});
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
    for (int qb = 0; qb < 4; ++qb) a.cand_cnt[(wave * 64 + qb * 16 + col) * nbuf + blockIdx.x * 4 + g] = st.cnt[qb];
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

  frag q[2][KS2];
  {
    const T* xq = (const T*)a.xq;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const uint32_t qi = wave * 32 + qb * 16 + col;
      const T* p = xq + (size_t)(qi < a.nq ? qi : a.nq - 1) * D + 8 * g;
#pragma unroll
      for (int s2 = 0; s2 < KS2; ++s2) {
        if (qb * KS2 + s2 < NAQ) agpr_load_frag(q[qb][s2], p + 32 * s2);
        else q[qb][s2] = *(const frag*)(p + 32 * s2);
      }
    }
#pragma unroll
    for (int i = 0; i < NAQ; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+a"(q[i / KS2][i % KS2]));
  }
  float thr[2];
  uint32_t cnt[2], off[2];
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
    const uint32_t tile = a.tile_first + j * a.tile_stride;
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
    const uint32_t row0 = (a.tile_first + j * a.tile_stride) * kTileRows + (u & 1) * 16 + 4 * g;
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
        const float m0 = max4v(acc[0]), m1 = max4v(acc[1]);
        if (__builtin_amdgcn_ballot_w64(m0 > thr[0] || m1 > thr[1])) {
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t id = row0 + i;
              if (acc[qb][i] > thr[qb] && id < a.n_rows) {
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
                if (lane == L) { cnt[qb] = a.k; thr[qb] = key_score(kth); }
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

// ---- wide rows (d > 1536, any multiple of 128), hand-pipelined -------------------------------------------------------
// The transposed design of flat_scan16_kernel: there the QUERIES stay in registers and the corpus streams past; here rows
// are too wide for that, so the ACCUMULATORS stay (all 256 AGPRs: 256 rows x 64 queries per wave) and both operands stream
// in 64-wide K steps: corpus slab (NT 32-row tiles x 128 B = 32 KB) HBM -> LDS by LDS-DMA through a 3-slot ring (two K
// steps ahead), the wave's 64 queries x 128 B from L2 straight into registers one K step ahead (scalar base + lane offset
// loads, double-buffered; every CU re-reads the query block once per 256 rows: 1 B of L2 traffic per corpus byte).
// Per K step and wave: 32 A fragments (ring of 8 ds_read_b128) x 4 MFMA 16x16x32, 8 DMA pieces, 8 query loads.
// The LDS image of a slab is NT copies of flat_scan16_kernel's 4 KB k-group block, so addressing is shared.
// vmcnt queue at the top of step s (oldest first): DMA(s) | q(s), DMA(s+1) -> wait vmcnt(NT): only DMA(s+1) may be out.
// Accumulators are HARD-WIRED AGPRs a[4i .. 4i+3]: 64 tied "+a" operands (256 registers) defeat hipcc's allocator (it
// shuttled them through scratch and v_accvgpr_mov at every loop edge).  Every statement that touches them names all 256 as
// clobbers, so the compiler keeps nothing of its own in AGPRs and sizes the kernel's register file for them.
#define RR_ALL_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
template <typename T> struct Mfma16Fixed;
#define RR_MFMA16F(NAME, MNEMONIC, FRAG)                                                                             \
  template <> struct Mfma16Fixed<NAME> {                                                                             \
    template <int R, bool FIRST>                                                                                     \
    static __device__ __forceinline__ void run(FRAG a, FRAG b) {                                                     \
      if (FIRST) asm volatile(MNEMONIC " a[%2:%3], %0, %1, 0" ::"v"(a), "v"(b), "n"(R), "n"(R + 3) : RR_ALL_AGPRS);  \
      else asm volatile(MNEMONIC " a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(R), "n"(R + 3) : RR_ALL_AGPRS); \
    }                                                                                                                \
  };
RR_MFMA16F(_Float16, "v_mfma_f32_16x16x32_f16", f16x8)
RR_MFMA16F(__bf16, "v_mfma_f32_16x16x32_bf16", bf16x8)
#undef RR_MFMA16F
template <int R>
__device__ __forceinline__ f32x4 read_acc_fixed() {
  f32x4 v;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3) : RR_ALL_AGPRS);
  return v;
}

template <typename F>
__device__ __forceinline__ void query_load_frag(F& dst, uint32_t lane_off, const void* sbase, int imm) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_off), "s"(sbase), "n"(imm) : "memory");
}

// (Carrying the fragment ring across K steps - 4 slots, the barrier moved to fragment 24 of the step before - measured 3.53 vs
// 3.66 TB/s: no gain, like the same experiment on flat_scan16_kernel; not kept.)
template <int N, typename F>
__device__ __forceinline__ void vm_wait_tied8(F& r0, F& r1, F& r2, F& r3, F& r4, F& r5, F& r6, F& r7) {
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "n"(N) : "memory");
}
// DENSE launches (bootstrap sample, corpora <= 8192 rows) cover at most 256 tiles: with 8-tile groups only 32 workgroups would
// have work and each would still walk the whole K loop (112 us at d = 4096), so they use one tile per group.
template <typename T, bool DENSE, bool L2 = false>
__global__ __launch_bounds__(256, 1) void flat_scan_wide_kernel(const ScanArgs a, const int D) {
  typedef typename Mfma<T>::frag frag;
  constexpr int NT = DENSE ? 1 : 8;        // 32-row tiles per group
  constexpr int STEP_BYTES = NT * 4096;    // one K step of one group in LDS
  constexpr int NS = 3;
  constexpr int LEAD = NS - 1;             // K steps the DMA stream runs ahead
  constexpr int NF = 4 * NT;               // A fragments per K step: (tile, s2, rb)
  constexpr int NB = NF < 8 ? NF : 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, g = lane >> 4;
  const int KG = D / 64;                   // even (D is a multiple of 128)
  const uint32_t n_tiles = a.n_tiles;
  const uint32_t n_groups = (n_tiles + NT - 1) / NT;

  uint32_t roff[2];
  {
    const int rho = col & 7, p = col >> 3;
    const int f = ((rho >> 1) & 3) | (p << 2);
#pragma unroll
    for (int par = 0; par < 2; ++par) roff[par] = p * 1024 + rho * 128 + (((4 * par + g) ^ f) * 16);
  }
  // DMA side: this wave's piece of tile t = rows 8*wave .. +7, lane -> (row rho_w, 16-byte chunk c_w)
  const int rho_w = lane >> 3, sig = lane & 7;
  const int f_w = ((rho_w >> 1) & 3) | ((wave & 1) << 2);
  const int c_w = sig ^ f_w;
  const size_t row_bytes = (size_t)D * 2;
  auto issue_piece = [&](uint32_t grp, int kg, int slot, int t) {
    uint32_t j = grp * NT + t;
    j = j < n_tiles ? j : n_tiles - 1;
    uint32_t row = (a.tile_first + j * a.tile_stride) * kTileRows + wave * 8 + rho_w;
    row = row < a.n_rows ? row : a.n_rows - 1;
    const char* gp = (const char*)a.xb + (size_t)row * row_bytes + (size_t)kg * 128 + c_w * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                     (__attribute__((address_space(3))) void*)(smem + slot * STEP_BYTES + t * 4096 + wave * 1024), 16, 0, 2);
  };

  const int nb = __builtin_amdgcn_readfirstlane(
      (int)a.nq <= wave * 64 ? 0 : ((int)a.nq - wave * 64 >= 64 ? 4 : ((int)a.nq - wave * 64 + 15) / 16));
  uint32_t qoff[4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const uint32_t qi = wave * 64 + qb * 16 + col;
    qoff[qb] = (qi < a.nq ? qi : a.nq - 1) * (uint32_t)(D * 2) + 16 * g;
  }
  LaneState4 st;
  const uint32_t nbuf = gridDim.x * 4;
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const uint32_t qi = wave * 64 + qb * 16 + col;
    st.thr[qb] = DENSE ? 0.f : a.thr[qi];
    st.cnt[qb] = 0;
    st.off[qb] = (qi * nbuf + blockIdx.x * 4 + g) * (uint32_t)a.cap;
  }

  uint32_t grp = blockIdx.x;               // group being multiplied
  uint32_t dgrp = blockIdx.x;              // group / K step the DMA stream is at (two steps ahead)
  int dkg = 0, dslot = 0;
  auto dma_advance = [&]() {
    if (++dkg == KG) { dkg = 0; dgrp += gridDim.x; }
    if (++dslot == NS) dslot = 0;
  };
  auto mfma4 = [&](auto r_tag, auto first_tag, frag x, frag q0, frag q1, frag q2, frag q3) {
    constexpr int R = decltype(r_tag)::value;
    constexpr bool F1 = decltype(first_tag)::value;
    Mfma16Fixed<T>::template run<R, F1>(x, q0);
    Mfma16Fixed<T>::template run<R + 4, F1>(x, q1);
    Mfma16Fixed<T>::template run<R + 8, F1>(x, q2);
    Mfma16Fixed<T>::template run<R + 12, F1>(x, q3);
  };
  auto read_tile = [&](auto r_tag, f32x4 (&e)[2][4]) {
    constexpr int R = decltype(r_tag)::value;
    e[0][0] = read_acc_fixed<R>();      e[0][1] = read_acc_fixed<R + 4>();  e[0][2] = read_acc_fixed<R + 8>();  e[0][3] = read_acc_fixed<R + 12>();
    e[1][0] = read_acc_fixed<R + 16>(); e[1][1] = read_acc_fixed<R + 20>(); e[1][2] = read_acc_fixed<R + 24>(); e[1][3] = read_acc_fixed<R + 28>();
  };
  frag q[2][4][2];                         // [buffer][query block][k slice]
  // The query loads are asynchronous asm: their destination registers must have LANDED before any point where hipcc may
  // copy them (it inserts v_mov copies of loop-carried values at loop back-edges).  So the wait for the queries of step
  // s+1 closes step s, and the statement names the registers as in/out operands: every later use or copy follows it.
  auto queries_landed = [&](auto buf_tag) {
    constexpr int B = decltype(buf_tag)::value;
    vm_wait_tied8<NT>(q[B][0][0], q[B][0][1], q[B][1][0], q[B][1][1], q[B][2][0], q[B][2][1], q[B][3][0], q[B][3][1]);
  };
  auto load_queries = [&](int buf, int kg) {
    const char* sb = (const char*)a.xq + (size_t)kg * 128;
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      query_load_frag(q[buf][qb][0], qoff[qb], sb, 0);
      query_load_frag(q[buf][qb][1], qoff[qb], sb, 64);
    }
  };
  if (grp < n_groups) {
    // prologue, in the steady-state queue order: DMA(0) [, DMA(1)] | q(0), DMA(LEAD-1)
#pragma unroll
    for (int sidx = 0; sidx < LEAD - 1; ++sidx) {
#pragma unroll
      for (int t = 0; t < NT; ++t) issue_piece(dgrp, dkg, dslot, t);
      dma_advance();
    }
    load_queries(0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) issue_piece(dgrp, dkg, dslot, t);
    dma_advance();
    queries_landed(std::integral_constant<int, 0>{});   // only DMA(LEAD-1) may still be in flight
  }

  int slot = 0;   // accumulator of (tile t, row block rb, query block qb): a[16 (2t + rb) + 4 qb ..+3]
  while (grp < n_groups) {
    for (int kg = 0; kg < KG; kg += 2) {
      auto step = [&](auto buf_tag, auto first_tag, int kgs) {
        constexpr int P = decltype(buf_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        // (this step's queries and, older in the queue, its slab pieces were waited for when the previous step closed)
        __builtin_amdgcn_s_barrier();
        {
          int nkg = kgs + 1;
          if (nkg == KG) nkg = 0;          // the next group starts over on the same queries
          load_queries(1 - P, nkg);
        }
        if (nb > 0) {
          frag c[NB];
          const uint32_t ab0 = (uint32_t)(slot * STEP_BYTES) + roff[0], ab1 = (uint32_t)(slot * STEP_BYTES) + roff[1];
          // fragment f: tile f>>2, k slice (f>>1)&1, row block f&1
#pragma unroll
          for (int f = 0; f < NB; ++f) lds_read_frag(c[f], ((f >> 1) & 1) ? ab1 : ab0, (f >> 2) * 4096 + (f & 1) * 2048);
          static_for<NF>([&](auto fi) {
            constexpr int f = decltype(fi)::value;
            constexpr int t = f >> 2, par = (f >> 1) & 1, rb = f & 1;
            if (NF - f >= NB) lgkm_wait<NB - 1>();
            else if (NF - f == 7) lgkm_wait<6>();
            else if (NF - f == 6) lgkm_wait<5>();
            else if (NF - f == 5) lgkm_wait<4>();
            else if (NF - f == 4) lgkm_wait<3>();
            else if (NF - f == 3) lgkm_wait<2>();
            else if (NF - f == 2) lgkm_wait<1>();
            else lgkm_wait<0>();
            mfma4(std::integral_constant<int, 16 * (2 * t + rb)>{}, std::integral_constant<bool, FIRST && par == 0>{}, c[f % NB],
                  q[P][0][par], q[P][1][par], q[P][2][par], q[P][3][par]);
            if (f + NB < NF) {
              constexpr int fn = f + NB;
              lds_read_frag(c[f % NB], ((fn >> 1) & 1) ? ab1 : ab0, (fn >> 2) * 4096 + (fn & 1) * 2048);
            }
            if ((f & 3) == 1) issue_piece(dgrp, dkg, dslot, f >> 2);
          });
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t) issue_piece(dgrp, dkg, dslot, t);
        }
        dma_advance();
        if (++slot == NS) slot = 0;
        queries_landed(std::integral_constant<int, 1 - P>{});  // queue: ... q(s+1) | DMA(s+LEAD): NT pieces may remain
      };
      if (kg == 0) step(std::integral_constant<int, 0>{}, std::true_type{}, kg);
      else step(std::integral_constant<int, 0>{}, std::false_type{}, kg);
      step(std::integral_constant<int, 1>{}, std::false_type{}, kg + 1);
    }
    if (nb > 0) {
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      static_for<NT>([&](auto ti) {
        constexpr int t = decltype(ti)::value;
        const uint32_t j = grp * NT + t;
        if (j < n_tiles) {
          f32x4 e[2][4];
          read_tile(std::integral_constant<int, 32 * t>{}, e);
          if (L2) {  // rank by q.x - |x|^2/2: the lane's 2 x 4 rows of this tile (rows past the end are filtered by id later)
            const uint32_t r0 = (a.tile_first + j * a.tile_stride) * kTileRows + 4 * g;
            f32x4 h0, h1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              h0[i] = a.half_sqnorm[r0 + i < a.n_rows ? r0 + i : a.n_rows - 1];
              h1[i] = a.half_sqnorm[r0 + 16 + i < a.n_rows ? r0 + 16 + i : a.n_rows - 1];
            }
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
              e[0][qb] -= h0;
              e[1][qb] -= h1;
            }
          }
          tile_epilogue16<DENSE, 4>(a, st, e, j, lane, wave);
        }
      });
    }
    grp += gridDim.x;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
  if (!DENSE) {
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) a.cand_cnt[(wave * 64 + qb * 16 + col) * nbuf + blockIdx.x * 4 + g] = st.cnt[qb];
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

template <typename T>
static hipError_t launch_scan_wide(const ScanArgs& a, int D, bool dense, int grid, hipStream_t st) {
  if (D % 128 != 0) return hipErrorInvalidValue;
  const size_t lds = 3 * (dense ? 1 : 8) * 4096;
  hipError_t e;
#define RR_LAUNCH_W(DENSE_, L2_)                                                                                             \
  {                                                                                                                         \
    e = hipFuncSetAttribute((const void*)flat_scan_wide_kernel<T, DENSE_, L2_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                                                          \
    hipLaunchKernelGGL((flat_scan_wide_kernel<T, DENSE_, L2_>), dim3(grid), dim3(256), lds, st, a, D);                     \
  }
  if (a.half_sqnorm) { if (dense) RR_LAUNCH_W(true, true) else RR_LAUNCH_W(false, true) }
  else { if (dense) RR_LAUNCH_W(true, false) else RR_LAUNCH_W(false, false) }
#undef RR_LAUNCH_W
  return hipGetLastError();
}

int g_wide_min_queries = 129;  // 768 < D <= 1536: batches of at least this many queries take the wide-row kernel in one pass (RR_WIDE_MIN_QUERIES)
int g_scan_variant = 16;       // 16 = default; 15 = the same kernel without non-temporal DMA; other values: development kernels (RR_DEV_VARIANTS builds)
int g_generic_tall = 16;       // 16 = flat_scan_wide_kernel; 8 / 4 / 0: compiler-scheduled wide-row forms (RR_DEV_VARIANTS builds)

static void read_variant_env() {
  static const bool env_read = [] {
    if (const char* v = getenv("RR_SCAN_VARIANT")) g_scan_variant = atoi(v);
    if (const char* v = getenv("RR_GENERIC_TALL")) g_generic_tall = atoi(v);
    if (const char* v = getenv("RR_WIDE_MIN_QUERIES")) g_wide_min_queries = atoi(v);
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

// candidate buffers per (workgroup, query) of the kernel that will serve this dim: the production kernels all use 4 lane quarters
int scan_bufs_per_wg(int D, bool l2) {
  read_variant_env();
#ifdef RR_DEV_VARIANTS
  return l2 ? 4 : dev_scan_bufs_per_wg(D, g_scan_variant, g_generic_tall);  // (the development kernels have no L2 form)
#else
  (void)D; (void)l2;
  return 4;
#endif
}

template <typename T, int D, bool DENSE, bool NT = false, bool L2 = false>
static hipError_t launch_scan16(const ScanArgs& a, int grid, hipStream_t st) {
  const size_t lds = scan16_slots(D) * ((size_t)kTileRows * D * 2 + 256);  // ring + L2 norm slots
  hipError_t e = hipFuncSetAttribute((const void*)flat_scan16_kernel<T, D, DENSE, NT, L2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((flat_scan16_kernel<T, D, DENSE, NT, L2>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <typename T, int D>
static hipError_t launch_scan_t(const ScanArgs& a, bool dense, int grid, hipStream_t st) {
  const bool nt = (size_t)a.n_rows * D * 2 > kNtThresholdBytes && g_scan_variant != 15;
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
  if (dev_scan_handles(a, D, g_scan_variant, g_generic_tall)) return launch_dev_scan(a, dtype, D, dense, grid, st, g_scan_variant, g_generic_tall);
#endif
  const bool f16 = dtype == RR_DTYPE_F16;
  if (half_resident_dim(D) && (int)a.nq < g_wide_min_queries && !a.half_sqnorm)  // 768 < D <= 1536, small batch: 32 resident queries per wave
    return f16 ? launch_scan_half_resident<_Float16>(a, D, dense, grid, st) : launch_scan_half_resident<__bf16>(a, D, dense, grid, st);
  if (D > kMaxResidentDim)  // wide rows; at 768 < D <= 1536 one pass of this kernel beats two passes of the half-resident one
    return f16 ? launch_scan_wide<_Float16>(a, D, dense, grid, st) : launch_scan_wide<__bf16>(a, D, dense, grid, st);
  return f16 ? launch_scan_d<_Float16>(a, D, dense, grid, st) : launch_scan_d<__bf16>(a, D, dense, grid, st);
}

int scan_padded_dim(int d) {
  static const int dims[] = {128, 256, 384, 512, 640, 768, 896, 1024, 1280, 1536};  // query-resident instantiations
  for (int v : dims)
    if (d <= v) return v;
  if (d <= kMaxDim) return (d + 127) / 128 * 128;  // wide-row kernel
  return -1;
}

}  // namespace rr
