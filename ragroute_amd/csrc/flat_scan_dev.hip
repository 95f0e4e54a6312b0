// Development kernels of K1: measured alternatives to the production kernels of flat_scan.hip, kept so the A/B numbers in
// DESIGN.md can be reproduced.  Compiled only when the library is built with RR_DEV_VARIANTS=1 (-DRR_DEV_VARIANTS); selected
// at run time by RR_SCAN_VARIANT (D = 768 only) and RR_GENERIC_TALL (wide rows):
//   flat_scan_kernel         the query-resident design on the 32x32x16 MFMA shape: variant 1 (asm loop, 6.7 % slower: lower
//                            clock), 0 (compiler-scheduled builtin MFMA, the first version), 2 (stamped diagnostic),
//                            4..9 / 48 timing-only ablations (additionally -DRR_ABLATION_VARIANTS)
//   flat_scan16x8_kernel     variant 8: two waves per SIMD, 32 queries per wave (11 % slower)
//   flat_scan_generic_kernel variant 3 / RR_GENERIC_TALL=0: queries re-streamed from L2 per 64 rows, compiler-scheduled
//   flat_scan_tall_kernel    RR_GENERIC_TALL=4 / 8: the same with 128 / 256 rows per iteration
#include "flat_scan_common.h"

#ifdef RR_DEV_VARIANTS
namespace rr {

template <typename T, int D, bool DENSE, int VARIANT>
__global__ __launch_bounds__(256, 1) void flat_scan_kernel(const ScanArgs a) {
  typedef typename Mfma<T>::frag frag;
  constexpr int KS = D / 16;  // 16-wide k slices (one MFMA each per query block)
  constexpr int KG = D / 64;  // 64-wide k groups (one 1 KiB DMA piece per 8 rows)
  constexpr int TILE_BYTES = kTileRows * D * 2;
  constexpr int NA1 = KS < 14 ? KS : 14;  // block-1 query fragments that also live in AGPRs (4*(KS+NA1) <= 248)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const uint32_t q0i = wave * 64 + r, q1i = q0i + 32;

  // ---- resident queries (MFMA B operand: lane holds query r, k = 16 s + 8 h .. +7) ----------
  // Rows past nq are clamped to the last query (their thresholds are +inf / their dense rows unused).
  frag q0[KS], q1[KS];
  {
    const T* xq = (const T*)a.xq;
    const uint32_t r0 = q0i < a.nq ? q0i : a.nq - 1, r1 = q1i < a.nq ? q1i : a.nq - 1;
    const T* p0 = xq + (size_t)r0 * D + 8 * h;
    const T* p1 = xq + (size_t)r1 * D + 8 * h;
    if (VARIANT == 0) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        q0[s] = *(const frag*)(p0 + 16 * s);
        q1[s] = *(const frag*)(p1 + 16 * s);
      }
    } else {
      // Block 0 (and the first NA1 fragments of block 1) are loaded STRAIGHT INTO AGPRs, so the values are
      // accumulator-file class for their whole life and the MFMAs read them there without copies.
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        agpr_load_frag(q0[s], p0 + 16 * s);
        if (s < NA1) agpr_load_frag(q1[s], p1 + 16 * s);
        else q1[s] = *(const frag*)(p1 + 16 * s);
      }
      // one wait that names every asm-loaded destination, before any consumer (hipcc does not count asm loads)
#pragma unroll
      for (int s = 0; s < KS; s += 8) {
        if (s + 8 <= KS)
          asm volatile("s_waitcnt vmcnt(0)" : "+a"(q0[s]), "+a"(q0[s + 1]), "+a"(q0[s + 2]), "+a"(q0[s + 3]), "+a"(q0[s + 4]),
                       "+a"(q0[s + 5]), "+a"(q0[s + 6]), "+a"(q0[s + 7]));
        else
          for (int t = s; t < KS; ++t) asm volatile("s_waitcnt vmcnt(0)" : "+a"(q0[t]));
      }
#pragma unroll
      for (int s = 0; s < NA1; ++s) asm volatile("s_waitcnt vmcnt(0)" : "+a"(q1[s]));
    }
  }

  LaneState st = {0.f, 0.f, 0, 0, 0, 0};
  const uint32_t nbuf = gridDim.x * 2;
  if (!DENSE) {
    st.thr0 = a.thr[q0i];
    st.thr1 = a.thr[q1i];
    st.off0 = (q0i * nbuf + blockIdx.x * 2 + h) * (uint32_t)a.cap;
    st.off1 = (q1i * nbuf + blockIdx.x * 2 + h) * (uint32_t)a.cap;
  }

  // ---- LDS image addressing -------------------------------------------------------------------
  // piece (kg, p) = rows 8p..8p+7, halves 64kg..64kg+63, at byte (kg*4+p)*1024; inside it the
  // 16-byte chunk c of row rho sits at rho*128 + (c ^ f(rho,p))*16, f = ((rho>>1)&3)|((p&1)<<2).
  const int p = r >> 3, rho = r & 7;
  const int f = ((rho >> 1) & 3) | ((p & 1) << 2);
  uint32_t roff[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) roff[s4] = p * 1024 + rho * 128 + (((2 * s4 + h) ^ f) * 16);
  // DMA side: wave w fills row group p = w; lane -> (rho_w, sigma) and fetches chunk sigma ^ f.
  const int rho_w = lane >> 3, sig = lane & 7;
  const int f_w = ((rho_w >> 1) & 3) | ((wave & 1) << 2);
  const int c_w = sig ^ f_w;

  // tile ordinal j -> per-lane global source address of this wave's row group (rows clamped into the corpus;
  // ordinals past the end re-load the last tile, which keeps the vmcnt bookkeeping uniform)
  auto tile_src = [&](uint32_t j) -> const char* {
    if (j >= a.n_tiles) j = a.n_tiles - 1;
    if (VARIANT == 48) j &= 1023;  // ablation: same DMA instructions, L2/MALL-resident source (48 MB)
    const uint32_t tile = a.tile_first + j * a.tile_stride;
    uint32_t row = tile * kTileRows + wave * 8 + rho_w;
    row = row < a.n_rows ? row : a.n_rows - 1;
    return (const char*)a.xb + (size_t)row * (D * 2) + c_w * 16;
  };
  auto issue_piece = [&](const char* g, int slot, int kg) {
    char* l = smem + slot * TILE_BYTES + wave * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + kg * 128),
                                     (__attribute__((address_space(3))) void*)(l + kg * 4096), 16, 0, 0);
  };

  uint32_t j = blockIdx.x;
  const uint32_t stride = gridDim.x;
  const uint32_t n_tiles = a.n_tiles;
  if (j < n_tiles) {
    const char* g0 = tile_src(j);
    const char* g1 = tile_src(j + stride);
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) issue_piece(g0, 0, kg);
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) issue_piece(g1, 1, kg);
  }
  int slot = 0;
  uint64_t seg0 = 0, seg1 = 0, seg2 = 0, seg3 = 0, tA = 0, tB = 0;  // VARIANT 2 only
  uint64_t c_begin = 0, r_begin = 0;
  if (VARIANT == 2) {
    c_begin = __builtin_amdgcn_s_memtime();
    r_begin = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
  for (; j < n_tiles; j += stride) {
    if (VARIANT == 2) tA = stamp();
    // tile j landed (this wave's pieces): all but the KG youngest DMA ops are done
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KG) : "memory");
    if (VARIANT == 2) { tB = stamp(); seg0 += tB - tA; tA = tB; }
    if (VARIANT != 7) __builtin_amdgcn_s_barrier();
    if (VARIANT == 2) { tB = stamp(); seg1 += tB - tA; tA = tB; }
    int nslot = slot + 2;
    if (nslot >= 3) nslot -= 3;
    const char* gn = tile_src(j + 2 * stride);

    f32x16 a0, a1;
    const char* base = smem + slot * TILE_BYTES;
    if (VARIANT == 0) {
      a0 = f32x16{0};
      a1 = f32x16{0};
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) issue_piece(gn, nslot, kg);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const frag c = *(const frag*)(base + roff[s & 3] + (s >> 2) * 4096);
        a0 = Mfma<T>::run(c, q0[s], a0);
        a1 = Mfma<T>::run(c, q1[s], a1);
      }
    } else {
      // software-pipelined by hand: NB corpus fragments in flight from LDS (counted lgkmcnt), one DMA piece of
      // the tile after next issued every 4 k-slices, MFMAs back to back
      constexpr int NB = KS < 8 ? KS : 8;
      frag c[NB];
      uint32_t ab[4];
      f32x4 gacc[4] = {{0}, {0}, {0}, {0}};  // VARIANT 9 only
#pragma unroll
      for (int i = 0; i < 4; ++i) ab[i] = (uint32_t)(slot * TILE_BYTES) + roff[i];
#pragma unroll
      for (int i = 0; i < NB; ++i) lds_read_frag(c[i], ab[i & 3], (i >> 2) * 4096);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        // reads outstanding now: min(NB, KS - s); the oldest is fragment s
        if (KS - s >= NB) lgkm_wait<NB - 1>();
        else if (KS - s == 7) lgkm_wait<6>();
        else if (KS - s == 6) lgkm_wait<5>();
        else if (KS - s == 5) lgkm_wait<4>();
        else if (KS - s == 4) lgkm_wait<3>();
        else if (KS - s == 3) lgkm_wait<2>();
        else if (KS - s == 2) lgkm_wait<1>();
        else lgkm_wait<0>();
        if (VARIANT == 9) {  // ablation: same operands and flops as 2 x 32x32x16, issued as 4 x 16x16x32 (timing only)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(gacc[0]) : "v"(c[s % NB]), "a"(q0[s]));
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(gacc[1]) : "v"(c[s % NB]), "a"(q0[s]));
          if (s < NA1) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(gacc[2]) : "v"(c[s % NB]), "a"(q1[s]));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(gacc[3]) : "v"(c[s % NB]), "a"(q1[s]));
          } else {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(gacc[2]) : "v"(c[s % NB]), "v"(q1[s]));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(gacc[3]) : "v"(c[s % NB]), "v"(q1[s]));
          }
        } else if (s == 0) {  // srcC = inline 0: no accumulator zero-fill
          MfmaAsm<T>::first_a(a0, c[0], q0[0]);
          MfmaAsm<T>::first_a(a1, c[0], q1[0]);
        } else {
          MfmaAsm<T>::acc_a(a0, c[s % NB], q0[s]);
          if (s < NA1) MfmaAsm<T>::acc_a(a1, c[s % NB], q1[s]);
          else MfmaAsm<T>::acc_v(a1, c[s % NB], q1[s]);
        }
        if (s + NB < KS && VARIANT != 6) lds_read_frag(c[s % NB], ab[(s + NB) & 3], ((s + NB) >> 2) * 4096);
        if ((s & 3) == 1 && VARIANT != 4) issue_piece(gn, nslot, s >> 2);
      }
      if (VARIANT == 9) { asm volatile("" ::"v"(gacc[0]), "v"(gacc[1]), "v"(gacc[2]), "v"(gacc[3])); a0 = f32x16{0}; a1 = f32x16{0}; }
      mfma_drain(a0, a1);
    }
    if (VARIANT == 2) { tB = stamp(); seg2 += tB - tA; tA = tB; }

    if (VARIANT != 5 && VARIANT != 9) tile_epilogue<DENSE>(a, st, a0, a1, j, q0i, q1i, h, lane, wave);
    else asm volatile("" ::"v"(a0), "v"(a1));
    slot = slot + 1;
    if (slot >= 3) slot = 0;
    if (VARIANT == 2) { tB = stamp(); seg3 += tB - tA; }
  }
  if (VARIANT == 2 && !DENSE && lane == 0) {  // stamps leave through the (otherwise unused) dense buffer only
    uint64_t* dbg = (uint64_t*)a.dense + (size_t)(blockIdx.x * 4 + wave) * 6;
    dbg[0] = seg0; dbg[1] = seg1; dbg[2] = seg2; dbg[3] = seg3;
    dbg[4] = __builtin_amdgcn_s_memtime() - c_begin;
    dbg[5] = __builtin_amdgcn_s_memrealtime() - r_begin;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
  if (!DENSE) {
    a.cand_cnt[q0i * nbuf + blockIdx.x * 2 + h] = st.cnt0;
    a.cand_cnt[q1i * nbuf + blockIdx.x * 2 + h] = st.cnt1;
  }
}

// =====================================================================================================
// ---- 8-wave form: two waves per SIMD, 32 resident queries per wave ---------------------------------------------
// One wave per SIMD cannot hide its own in-order stalls: every LDS-DMA issue holds the wave ~50 cycles past the MFMA's
// free issue slots, and the epilogue and the first fragment reads of a tile leave the matrix pipe idle.  With two
// waves per SIMD the partner's MFMAs fill those holes.  Cost: 256 registers per lane (192 query + 16 accumulator +
// 16 fragment ring + ~30), and every tile is read from LDS by 8 waves instead of 4.
template <typename T, int D, bool DENSE>
__global__ __launch_bounds__(512, 2) void flat_scan16x8_kernel(const ScanArgs a) {
  typedef typename Mfma<T>::frag frag;
  constexpr int KS2 = D / 32, KG = D / 64, NF = 2 * KS2;
  constexpr int NQ = 2 * KS2;                 // resident query fragments per wave
  constexpr int NAQ = NQ < 32 ? NQ : 32;      // of which in AGPRs
  constexpr int TILE_BYTES = kTileRows * D * 2;
  constexpr int PIECES = KG / 2;              // DMA pieces per wave and tile (KG is even: D is a multiple of 128)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0..7
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

  LaneState4 st;
  lane_state_segments_init(a, st);
  const uint32_t nbuf = gridDim.x * 4;
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const uint32_t qi = wave * 32 + (qb & 1) * 16 + col;
    st.thr[qb] = (DENSE || qb >= 2) ? 0.f : a.thr[qi];
    st.cnt[qb] = 0;
    st.off[qb] = (qi * nbuf + blockIdx.x * 4 + g) * (uint32_t)a.cap;
  }

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
  // DMA: wave w fills row group p = w&3 of the k groups with parity w>>2
  const int pw = wave & 3, kpar = wave >> 2;
  const int rho_w = lane >> 3, sig = lane & 7;
  const int f_w = ((rho_w >> 1) & 3) | ((pw & 1) << 2);
  const int c_w = sig ^ f_w;
  auto tile_src = [&](uint32_t j) -> const char* {
    if (j >= a.n_tiles) j = a.n_tiles - 1;
    const uint32_t tile = a.tile_first + j * a.tile_stride;
    uint32_t row = tile * kTileRows + pw * 8 + rho_w;
    row = row < a.n_rows ? row : a.n_rows - 1;
    return (const char*)a.xb + (size_t)row * (D * 2) + c_w * 16 + kpar * 128;
  };
  auto issue_piece = [&](const char* gp, int slot, int i) {  // i-th piece of this wave: k group kpar + 2 i
    char* l = smem + slot * TILE_BYTES + pw * 1024 + kpar * 4096;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + i * 256),
                                     (__attribute__((address_space(3))) void*)(l + i * 8192), 16, 0, 0);
  };
  const int nb = __builtin_amdgcn_readfirstlane(
      (int)a.nq <= wave * 32 ? 0 : ((int)a.nq - wave * 32 >= 32 ? 2 : ((int)a.nq - wave * 32 + 15) / 16));

  uint32_t j = blockIdx.x;
  const uint32_t stride = gridDim.x;
  const uint32_t n_tiles = a.n_tiles;
  if (j < n_tiles) {
    const char* g0 = tile_src(j);
    const char* g1 = tile_src(j + stride);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) issue_piece(g0, 0, i);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) issue_piece(g1, 1, i);
  }
  int slot = 0;
  for (; j < n_tiles; j += stride) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    __builtin_amdgcn_s_barrier();
    int nslot = slot + 2;
    if (nslot >= 3) nslot -= 3;
    const char* gn = tile_src(j + 2 * stride);
    auto compute = [&](auto tag) {
      constexpr int NQB = decltype(tag)::value;
      f32x4 acc[2][4];
      constexpr int NB = NF < 4 ? NF : 4;
      frag c[NB];
      uint32_t ab[2][2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int par = 0; par < 2; ++par) ab[rb][par] = (uint32_t)(slot * TILE_BYTES) + roff[rb][par];
#pragma unroll
      for (int f = 0; f < NB; ++f) lds_read_frag(c[f], ab[f & 1][(f >> 1) & 1], (f >> 2) * 4096);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int s2 = f >> 1, rb = f & 1;
        if (NF - f >= NB) lgkm_wait<NB - 1>();
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
        if ((f & 7) == 3 && (f >> 3) < PIECES) issue_piece(gn, nslot, f >> 3);
      }
      if (NQB == 2) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
      else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][0]));
      tile_epilogue16<DENSE, NQB, 32>(a, st, acc, j, lane, wave);
    };
    if (nb == 2) {
      compute(std::integral_constant<int, 2>{});
    } else if (nb == 1) {
      compute(std::integral_constant<int, 1>{});
    } else {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) issue_piece(gn, nslot, i);
    }
    slot = slot + 1;
    if (slot >= 3) slot = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!DENSE) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) a.cand_cnt[(wave * 32 + qb * 16 + col) * nbuf + blockIdx.x * 4 + g] = st.cnt[qb];
  }
}

// ---- generic embedding dimension (any multiple of 64, e.g. 1024 / 4096 of FeB4RAG, config.py:45-57) ------------
// Queries no longer fit the register file, so they are re-streamed from L2 per 64-wide K step (each wave loads only
// its own 64 queries: nothing to share, no LDS hop); the corpus goes global -> registers -> LDS (same XOR-swizzled
// piece layout as the fast kernel) and is shared by the 4 waves.  Two 32-row tiles per iteration halve the L2
// traffic of the queries.  Compiler-scheduled (builtin MFMA, __syncthreads); same epilogue, same host schedule.
template <typename T, bool DENSE>
__global__ __launch_bounds__(256, 2) void flat_scan_generic_kernel(const ScanArgs a, const int D) {
  typedef typename Mfma<T>::frag frag;
  __shared__ __attribute__((aligned(16))) char smem[2][2 * 4096];  // [buffer][tile A | tile B] 32 rows x 128 B each
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const uint32_t q0i = wave * 64 + r, q1i = q0i + 32;
  LaneState st = {0.f, 0.f, 0, 0, 0, 0};
  const uint32_t nbuf = gridDim.x * 2;
  if (!DENSE) {
    st.thr0 = a.thr[q0i];
    st.thr1 = a.thr[q1i];
    st.off0 = (q0i * nbuf + blockIdx.x * 2 + h) * (uint32_t)a.cap;
    st.off1 = (q1i * nbuf + blockIdx.x * 2 + h) * (uint32_t)a.cap;
  }
  const T* xq = (const T*)a.xq;
  const T* p0 = xq + (size_t)(q0i < a.nq ? q0i : a.nq - 1) * D + 8 * h;
  const T* p1 = xq + (size_t)(q1i < a.nq ? q1i : a.nq - 1) * D + 8 * h;
  // read side (MFMA A fragment of slice s: row r, chunk 2s+h)
  uint32_t roff[4];
  {
    const int p = r >> 3, rho = r & 7, f = ((rho >> 1) & 3) | ((p & 1) << 2);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) roff[s4] = p * 1024 + rho * 128 + (((2 * s4 + h) ^ f) * 16);
  }
  // write side: thread -> (tile A/B, row, chunk pair)
  const int tsel = tid >> 7, wrow = (tid >> 2) & 31, cp = tid & 3;
  uint32_t woff0, woff1;
  {
    const int p = wrow >> 3, rho = wrow & 7, f = ((rho >> 1) & 3) | ((p & 1) << 2);
    woff0 = tsel * 4096 + p * 1024 + rho * 128 + (((2 * cp) ^ f) * 16);
    woff1 = tsel * 4096 + p * 1024 + rho * 128 + (((2 * cp + 1) ^ f) * 16);
  }
  const int KG = D / 64;
  const uint32_t n_tiles = a.n_tiles;
  for (uint32_t j = 2 * blockIdx.x; j < n_tiles; j += 2 * gridDim.x) {
    const bool validB = j + 1 < n_tiles;
    const uint32_t jsel = tsel ? (validB ? j + 1 : j) : j;
    uint32_t row = (a.tile_first + jsel * a.tile_stride) * kTileRows + wrow;
    row = row < a.n_rows ? row : a.n_rows - 1;
    const char* g = (const char*)a.xb + (size_t)row * ((size_t)D * 2) + cp * 32;
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    uint4 cr0 = *(const uint4*)g, cr1 = *(const uint4*)(g + 16);
    frag qa[4], qb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qa[s] = *(const frag*)(p0 + 16 * s);
      qb[s] = *(const frag*)(p1 + 16 * s);
    }
    for (int kg = 0; kg < KG; ++kg) {
      char* buf = smem[kg & 1];
      *(uint4*)(buf + woff0) = cr0;
      *(uint4*)(buf + woff1) = cr1;
      frag ca[4], cb[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { ca[s] = qa[s]; cb[s] = qb[s]; }
      __syncthreads();
      if (kg + 1 < KG) {
        cr0 = *(const uint4*)(g + (size_t)(kg + 1) * 128);
        cr1 = *(const uint4*)(g + (size_t)(kg + 1) * 128 + 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          qa[s] = *(const frag*)(p0 + (kg + 1) * 64 + 16 * s);
          qb[s] = *(const frag*)(p1 + (kg + 1) * 64 + 16 * s);
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const frag xA = *(const frag*)(buf + roff[s]);
        const frag xB = *(const frag*)(buf + 4096 + roff[s]);
        acc00 = Mfma<T>::run(xA, ca[s], acc00);
        acc01 = Mfma<T>::run(xA, cb[s], acc01);
        acc10 = Mfma<T>::run(xB, ca[s], acc10);
        acc11 = Mfma<T>::run(xB, cb[s], acc11);
      }
    }
    __syncthreads();  // the next pair restages buffer 0
    tile_epilogue<DENSE>(a, st, acc00, acc01, j, q0i, q1i, h, lane, wave);
    if (validB) tile_epilogue<DENSE>(a, st, acc10, acc11, j + 1, q0i, q1i, h, lane, wave);
  }
  if (!DENSE) {
    a.cand_cnt[q0i * nbuf + blockIdx.x * 2 + h] = st.cnt0;
    a.cand_cnt[q1i * nbuf + blockIdx.x * 2 + h] = st.cnt1;
  }
}

// ---- generic dimension, tall tiles -----------------------------------------------------------------------------------
// Same data flow as flat_scan_generic_kernel, but NT 32-row tiles (256 rows) per iteration: the queries a wave re-streams
// from L2 per 64-wide K step (8 KB) are amortised over 4x the rows, so L2 traffic per corpus byte drops from 4 to 1 and
// the 128 MFMAs of a K step cover its loads.  16 accumulators x 16 registers live in AGPRs; one workgroup per CU.
// Staging: one load instruction of the workgroup covers one 32-row tile slab (8 lanes per 128-byte row slice).
template <typename T, bool DENSE, int NT>
__global__ __launch_bounds__(256, 1) void flat_scan_tall_kernel(const ScanArgs a, const int D) {
  typedef typename Mfma<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][NT tiles][32 rows x 128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const uint32_t q0i = wave * 64 + r, q1i = q0i + 32;
  LaneState st = {0.f, 0.f, 0, 0, 0, 0};
  const uint32_t nbuf = gridDim.x * 2;
  if (!DENSE) {
    st.thr0 = a.thr[q0i];
    st.thr1 = a.thr[q1i];
    st.off0 = (q0i * nbuf + blockIdx.x * 2 + h) * (uint32_t)a.cap;
    st.off1 = (q1i * nbuf + blockIdx.x * 2 + h) * (uint32_t)a.cap;
  }
  const T* xq = (const T*)a.xq;
  const T* p0 = xq + (size_t)(q0i < a.nq ? q0i : a.nq - 1) * D + 8 * h;
  const T* p1 = xq + (size_t)(q1i < a.nq ? q1i : a.nq - 1) * D + 8 * h;
  uint32_t roff[4];
  {
    const int p = r >> 3, rho = r & 7, f = ((rho >> 1) & 3) | ((p & 1) << 2);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) roff[s4] = p * 1024 + rho * 128 + (((2 * s4 + h) ^ f) * 16);
  }
  const int wrow = tid >> 3, wc = tid & 7;
  uint32_t woff;
  {
    const int p = wrow >> 3, rho = wrow & 7, f = ((rho >> 1) & 3) | ((p & 1) << 2);
    woff = p * 1024 + rho * 128 + ((wc ^ f) * 16);
  }
  const int KG = D / 64;
  const uint32_t n_tiles = a.n_tiles;
  for (uint32_t j = NT * blockIdx.x; j < n_tiles; j += NT * gridDim.x) {
    const char* g[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const uint32_t jt = j + t < n_tiles ? j + t : n_tiles - 1;
      uint32_t row = (a.tile_first + jt * a.tile_stride) * kTileRows + wrow;
      row = row < a.n_rows ? row : a.n_rows - 1;
      g[t] = (const char*)a.xb + (size_t)row * ((size_t)D * 2) + wc * 16;
    }
    f32x16 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t][0] = f32x16{0};
      acc[t][1] = f32x16{0};
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 cr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cr[t] = *(const u32x4*)g[t];
    frag qa[4], qb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qa[s] = *(const frag*)(p0 + 16 * s);
      qb[s] = *(const frag*)(p1 + 16 * s);
    }
    for (int kg = 0; kg < KG; ++kg) {
      char* buf = smem + (kg & 1) * (NT * 4096);
#pragma unroll
      for (int t = 0; t < NT; ++t) *(u32x4*)(buf + t * 4096 + woff) = cr[t];
      frag ca[4], cb[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { ca[s] = qa[s]; cb[s] = qb[s]; }
      __syncthreads();
      if (kg + 1 < KG) {
#pragma unroll
        for (int t = 0; t < NT; ++t) cr[t] = *(const u32x4*)(g[t] + (size_t)(kg + 1) * 128);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          qa[s] = *(const frag*)(p0 + (kg + 1) * 64 + 16 * s);
          qb[s] = *(const frag*)(p1 + (kg + 1) * 64 + 16 * s);
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const frag x = *(const frag*)(buf + t * 4096 + roff[s]);
          acc[t][0] = Mfma<T>::run(x, ca[s], acc[t][0]);
          acc[t][1] = Mfma<T>::run(x, cb[s], acc[t][1]);
        }
      }
    }
    __syncthreads();  // the next group restages buffer 0
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (j + t < n_tiles) tile_epilogue<DENSE>(a, st, acc[t][0], acc[t][1], j + t, q0i, q1i, h, lane, wave);
  }
  if (!DENSE) {
    a.cand_cnt[q0i * nbuf + blockIdx.x * 2 + h] = st.cnt0;
    a.cand_cnt[q1i * nbuf + blockIdx.x * 2 + h] = st.cnt1;
  }
}

template <typename T, int D, bool DENSE, int VARIANT>
static hipError_t launch_scan_v(const ScanArgs& a, int grid, hipStream_t st) {
  const size_t lds = 3 * (size_t)kTileRows * D * 2;
  hipError_t e = hipFuncSetAttribute((const void*)flat_scan_kernel<T, D, DENSE, VARIANT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((flat_scan_kernel<T, D, DENSE, VARIANT>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <typename T, int D, bool DENSE>
static hipError_t launch_scan16x8(const ScanArgs& a, int grid, hipStream_t st) {
  const size_t lds = 3 * (size_t)kTileRows * D * 2;
  hipError_t e = hipFuncSetAttribute((const void*)flat_scan16x8_kernel<T, D, DENSE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((flat_scan16x8_kernel<T, D, DENSE>), dim3(grid), dim3(512), lds, st, a);
  return hipGetLastError();
}

template <typename T, int NT>
static hipError_t launch_scan_tall(const ScanArgs& a, int D, bool dense, int grid, hipStream_t st) {
  const size_t lds = 2 * NT * 4096;
  hipError_t e;
  if (dense) {
    e = hipFuncSetAttribute((const void*)flat_scan_tall_kernel<T, true, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((flat_scan_tall_kernel<T, true, NT>), dim3(grid), dim3(256), lds, st, a, D);
  } else {
    e = hipFuncSetAttribute((const void*)flat_scan_tall_kernel<T, false, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((flat_scan_tall_kernel<T, false, NT>), dim3(grid), dim3(256), lds, st, a, D);
  }
  return hipGetLastError();
}
template <typename T>
static hipError_t launch_scan_generic_dev(const ScanArgs& a, int D, bool dense, int grid, hipStream_t st, int tall) {
  if (D % 64 != 0 || a.half_sqnorm) return hipErrorNotSupported;
  if (tall == 8) return launch_scan_tall<T, 8>(a, D, dense, grid, st);
  if (tall) return launch_scan_tall<T, 4>(a, D, dense, grid, st);
  // two workgroups per CU (<= 256 registers per lane, 16 KB LDS): thread-level parallelism hides the L2 / barrier latency
  if (dense) hipLaunchKernelGGL((flat_scan_generic_kernel<T, true>), dim3(2 * grid), dim3(256), 0, st, a, D);
  else hipLaunchKernelGGL((flat_scan_generic_kernel<T, false>), dim3(2 * grid), dim3(256), 0, st, a, D);
  return hipGetLastError();
}

template <typename T> constexpr bool dtype_is_f16() { return false; }
template <> constexpr bool dtype_is_f16<_Float16>() { return true; }

static bool variant_768(int v) { return v == 0 || v == 1 || v == 2 || v == 8 || (v >= 4 && v <= 7) || v == 9 || v == 48; }
static bool generic_selected(int D, int variant, int tall) { return variant == 3 || (D > kMaxResidentDim && tall != 16); }

bool dev_scan_handles(const ScanArgs& a, int D, int variant, int tall) {
  if (a.half_sqnorm) return false;
  return (D == 768 && variant_768(variant)) || generic_selected(D, variant, tall);
}

int dev_scan_bufs_per_wg(int D, int variant, int tall) {
  if (generic_selected(D, variant, tall)) return (tall == 4 || tall == 8) ? 2 : 4;  // tall: 2 lane halves; 64-row form: 2 workgroups per CU x 2
  if (D == 768 && variant_768(variant) && variant != 8) return 2;                    // 32x32x16 shape: 2 lane halves
  return 4;
}

template <typename T>
static hipError_t launch_dev_t(const ScanArgs& a, int D, bool dense, int grid, hipStream_t st, int variant, int tall) {
  if (generic_selected(D, variant, tall)) return launch_scan_generic_dev<T>(a, D, dense, grid, st, tall == 16 ? 0 : tall);
  constexpr int DV = 768;
  if (variant == 8) return dense ? launch_scan16x8<T, DV, true>(a, grid, st) : launch_scan16x8<T, DV, false>(a, grid, st);
  if (variant == 0) return dense ? launch_scan_v<T, DV, true, 0>(a, grid, st) : launch_scan_v<T, DV, false, 0>(a, grid, st);
  if (variant == 1) return dense ? launch_scan_v<T, DV, true, 1>(a, grid, st) : launch_scan_v<T, DV, false, 1>(a, grid, st);
  if (variant == 2 && !dense) return launch_scan_v<T, DV, false, 2>(a, grid, st);
  if (variant == 2) return launch_scan_v<T, DV, true, 1>(a, grid, st);
#ifdef RR_ABLATION_VARIANTS
  if (!dense) {
    if (variant == 4) return launch_scan_v<T, DV, false, 4>(a, grid, st);
    if (variant == 5) return launch_scan_v<T, DV, false, 5>(a, grid, st);
    if (variant == 6) return launch_scan_v<T, DV, false, 6>(a, grid, st);
    if (variant == 7) return launch_scan_v<T, DV, false, 7>(a, grid, st);
    if (variant == 48) return launch_scan_v<T, DV, false, 48>(a, grid, st);
    if (variant == 9 && dtype_is_f16<T>()) return launch_scan_v<T, DV, false, 9>(a, grid, st);
  }
  if (variant >= 4) return launch_scan_v<T, DV, true, 1>(a, grid, st);
#endif
  return hipErrorNotSupported;
}

hipError_t launch_dev_scan(const ScanArgs& a, int dtype, int D, bool dense, int grid, hipStream_t st, int variant, int tall) {
  if (dtype == RR_DTYPE_F16) return launch_dev_t<_Float16>(a, D, dense, grid, st, variant, tall);
  if (dtype == RR_DTYPE_BF16) return launch_dev_t<__bf16>(a, D, dense, grid, st, variant, tall);
  return hipErrorInvalidValue;
}

}  // namespace rr
#endif  // RR_DEV_VARIANTS
