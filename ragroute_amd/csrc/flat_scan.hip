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
//  * Per tile: 48 x { ds_read_b128 corpus fragment ; 4 x v_mfma_f32_16x16x32 } (default shape; 2 x 32x32x16 in the
//    older form) with the corpus as A and the queries as B, so every lane ends up owning ONE query (column) and a
//    few corpus rows per result block: the top-k filter is a per-lane compare against that query's threshold, no
//    cross-lane work.  No inter-workgroup reuse exists (every byte is read once), so no XCD-aware block remap.
//  * Scores strictly above the threshold are appended (as 64-bit order keys) to a private
//    per-(workgroup, query, lane-quarter) buffer; if a buffer fills, the wave compacts it exactly
//    to its k best and raises that lane's threshold.  Nothing is ever dropped that could be in
//    the final top-k (see DESIGN.md "exactness").
//  * DENSE=true writes every score instead (bootstrap sample and tiny corpora).
//
// Kernels in this file (launch_flat_scan picks one; RR_SCAN_VARIANT overrides for A/B runs at D = 768):
//   flat_scan16_kernel      default, D <= 768: 16x16x32 MFMA, hand-pipelined asm loop, nt LDS-DMA, optional L2 metric
//   flat_scan16h_kernel     768 < D <= 1536: 32 resident queries per wave, half-tile LDS ring (128 queries per launch)
//   flat_scan_generic_kernel  D <= 8192 (multiples of 64): queries re-streamed from L2, compiler-scheduled  (variant 3)
//   flat_scan_kernel        the same design on the 32x32x16 shape: variant 1 (asm loop, 6.7 % slower: lower clock),
//                           variant 0 (compiler-scheduled builtin MFMA, the first version), variant 2 (stamped
//                           diagnostic), 4..9 / 48 timing-only ablations (-DRR_ABLATION_VARIANTS)
//   flat_scan16x8_kernel    variant 8: two waves per SIMD, 32 queries per wave (11 % slower; kept as a measured negative)
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "rr_common.h"
#include "rr_kernels.h"

namespace rr {

template <typename T> struct Mfma;
template <> struct Mfma<_Float16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma<__bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// int8 screening copy (rr_flat_search_screened): rows are viewed as pairs of bytes, so a row of D "elements" is 2*D
// int8 values and every address computation below is the f16 one; only the MFMA opcode (K = 64 bytes per 16-byte lane
// operand) and the accumulator type differ.  Only flat_scan16_kernel is instantiated for it.
struct I8Pair { int16_t v; };
template <> struct Mfma<I8Pair> { typedef s16x8 frag; };

// Inline-asm MFMA: accumulator in VGPRs (the epilogue's VALU reads it there), corpus fragment (A) in VGPRs, query fragment (B) either in AGPRs
// (block 0) or VGPRs (block 1).  hipcc otherwise keeps part of the resident queries in AGPRs and copies
// them to VGPRs with v_accvgpr_read before every MFMA (~250 copies per tile).  The accumulate chain
// (srcC == vDst) needs no wait states; the reader after the chain is fenced by mfma_drain().
template <typename T> struct MfmaAsm;
template <> struct MfmaAsm<_Float16> {
  static __device__ __forceinline__ void first_a(f32x16& c, f16x8 a, f16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "a"(b));
  }
  static __device__ __forceinline__ void acc_a(f32x16& c, f16x8 a, f16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
  }
  static __device__ __forceinline__ void acc_v(f32x16& c, f16x8 a, f16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
};
template <> struct MfmaAsm<__bf16> {
  static __device__ __forceinline__ void first_a(f32x16& c, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "a"(b));
  }
  static __device__ __forceinline__ void acc_a(f32x16& c, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
  }
  static __device__ __forceinline__ void acc_v(f32x16& c, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
};
// Global load of one fragment directly into accumulator registers (gfx90a+ VMEM may target AGPRs).
template <typename F, typename P>
__device__ __forceinline__ void agpr_load_frag(F& dst, const P* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(dst) : "v"(ptr) : "memory");
}
// LDS fragment read and counted wait, hidden from hipcc's waitcnt pass on purpose: it answers an asm consumer
// with lgkmcnt(0), which drains the whole fragment ring.  LDS reads of one wave return in order, so
// lgkmcnt(N) = "all but the N youngest reads have landed".
template <typename F>
__device__ __forceinline__ void lds_read_frag(F& dst, uint32_t addr, int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
}
// Diagnostic build only (VARIANT 2): shader-clock stamp, fenced so the segment it closes is complete.
__device__ __forceinline__ uint64_t stamp() {
  uint64_t t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
// 8-pass XDL result -> any non-MFMA reader needs 18 wait states the compiler cannot see inside asm.
__device__ __forceinline__ void mfma_drain(f32x16& a0, f32x16& a1) {
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a0), "+v"(a1));
}

// Wave-cooperative exact compaction of one lane's candidate buffer: keep the k largest keys
// (sorted, descending) and return the k-th key.  All 64 lanes participate; buf/scratch/cnt are
// wave-uniform.  Keys are unique (ids are unique), so ranks form a permutation.
__device__ __noinline__ uint64_t wave_compact(uint64_t* buf, uint64_t* scratch, int cnt, int k, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int e0 = 0; e0 < cnt; e0 += 64) {
    const int e = e0 + lane;
    const uint64_t mine = e < cnt ? buf[e] : ~0ull;
    int rank = 0;
    for (int j0 = 0; j0 < cnt; j0 += 64) {
      const uint64_t v = (j0 + lane) < cnt ? buf[j0 + lane] : 0ull;
      for (int t = 0; t < 64; ++t) {
        const uint64_t o = __shfl(v, t, 64);
        rank += o > mine ? 1 : 0;
      }
    }
    if (e < cnt && rank < k) scratch[rank] = mine;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int e = lane; e < k; e += 64) buf[e] = scratch[e];
  const uint64_t kth = scratch[k - 1];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return kth;
}

// max without the canonicalising v_max(x,x) hipcc puts in front of fmaxf on MFMA results (NaN operands lose, as maxNum)
__device__ __forceinline__ float max4(float a, float b, float c, float d) {
  float t, r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "v"(b), "v"(c));
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(d));
  return r;
}

// Per-lane filter state: strict thresholds, candidate counts and buffer offsets of the lane's two queries.
struct LaneState {
  float thr0, thr1;
  uint32_t cnt0, cnt1, off0, off1;
};

// Epilogue of one 32-row tile: a0/a1 hold the lane's 16 corpus rows (m = (i&3) + 8*(i>>2) + 4h) for its query of
// block 0 / block 1.  DENSE: store every score.  Otherwise: strict-threshold filter, append survivors as order keys
// to the lane's private buffers, compact exactly when a buffer is nearly full.
template <bool DENSE>
__device__ __forceinline__ void tile_epilogue(const ScanArgs& a, LaneState& st, const f32x16& a0, const f32x16& a1, uint32_t j,
                                              uint32_t q0i, uint32_t q1i, int h, int lane, int wave) {
    const uint32_t tile = a.tile_first + j * a.tile_stride;
    if (DENSE) {
      // column = j*32 + m, m = (i&3) + 8*(i>>2) + 4h
      float* d0 = a.dense + (size_t)q0i * a.dense_ld + j * kTileRows + 4 * h;
      float* d1 = a.dense + (size_t)q1i * a.dense_ld + j * kTileRows + 4 * h;
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        *(f32x4*)(d0 + 8 * i4) = f32x4{a0[4 * i4], a0[4 * i4 + 1], a0[4 * i4 + 2], a0[4 * i4 + 3]};
        *(f32x4*)(d1 + 8 * i4) = f32x4{a1[4 * i4], a1[4 * i4 + 1], a1[4 * i4 + 2], a1[4 * i4 + 3]};
      }
    } else {
      // two-level test: maxima of the four 4-register groups (rows 8g+4h .. +3), then their maximum
      float g0[4], g1[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        g0[g] = max4(a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3]);
        g1[g] = max4(a1[4 * g], a1[4 * g + 1], a1[4 * g + 2], a1[4 * g + 3]);
      }
      const float m0 = max4(g0[0], g0[1], g0[2], g0[3]);
      const float m1 = max4(g1[0], g1[1], g1[2], g1[3]);
      if (__builtin_amdgcn_ballot_w64(m0 > st.thr0 || m1 > st.thr1)) {
        // rare: a typical hit is ONE lane with ONE score, so only the group that holds it is expanded
        const uint32_t row0 = tile * kTileRows + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (__builtin_amdgcn_ballot_w64(g0[g] > st.thr0)) {
#pragma unroll
            for (int i = 4 * g; i < 4 * g + 4; ++i) {
              const uint32_t id = row0 + (i & 3) + 8 * (i >> 2);
              if (a0[i] > st.thr0 && id < a.n_rows) {
                a.cand[(size_t)st.off0 + st.cnt0] = make_key(a0[i], id);
                ++st.cnt0;
              }
            }
          }
          if (__builtin_amdgcn_ballot_w64(g1[g] > st.thr1)) {
#pragma unroll
            for (int i = 4 * g; i < 4 * g + 4; ++i) {
              const uint32_t id = row0 + (i & 3) + 8 * (i >> 2);
              if (a1[i] > st.thr1 && id < a.n_rows) {
                a.cand[(size_t)st.off1 + st.cnt1] = make_key(a1[i], id);
                ++st.cnt1;
              }
            }
          }
        }
        // keep >= 16 free slots per buffer; compaction is exact and raises the lane threshold
        const uint32_t lim = (uint32_t)a.cap - 16u;
        if (__builtin_amdgcn_ballot_w64(st.cnt0 > lim || st.cnt1 > lim)) {
          uint64_t* scratch = a.scratch + (size_t)(blockIdx.x * 4 + wave) * a.cap;
#pragma unroll 1
          for (int b = 0; b < 2; ++b) {
            uint64_t mask = __builtin_amdgcn_ballot_w64((b ? st.cnt1 : st.cnt0) > lim);
            while (mask) {
              const int L = __builtin_ctzll(mask);
              mask &= mask - 1;
              const uint32_t off = __shfl(b ? st.off1 : st.off0, L, 64);
              const int cnt = (int)__shfl(b ? st.cnt1 : st.cnt0, L, 64);
              const uint64_t kth = wave_compact(a.cand + (size_t)__builtin_amdgcn_readfirstlane(off), scratch,
                                                __builtin_amdgcn_readfirstlane(cnt), a.k, lane);
              if (lane == L) {
                if (b) { st.cnt1 = a.k; st.thr1 = key_score(kth); }
                else   { st.cnt0 = a.k; st.thr0 = key_score(kth); }
              }
            }
          }
        }
      }
    }
}

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
// 16x16x32 MFMA shape.  Same design, registers and LDS image as flat_scan_kernel; the chip holds a higher clock on
// this shape under the combined HBM + MFMA load (measured: -6.7 % time at equal flops and operands), which is what
// bounds the kernel.  Per wave: 4 query blocks of 16 (B operand) x 2 row blocks of 16 (A operand); D fragment:
// column = lane&15 -> query, rows 4*(lane>>4) + reg.  A lane therefore owns 4 queries x 8 rows per tile, and the
// candidate buffers are per (workgroup, query, lane quarter): 4 per workgroup and query.
template <typename T> struct Mfma16Asm;
#define RR_MFMA16(NAME, MNEMONIC, FRAG)                                                                        \
  template <> struct Mfma16Asm<NAME> {                                                                          \
    static __device__ __forceinline__ void first_a(f32x4& c, FRAG a, FRAG b) {                                  \
      asm volatile(MNEMONIC " %0, %1, %2, 0" : "=v"(c) : "v"(a), "a"(b));                                      \
    }                                                                                                           \
    static __device__ __forceinline__ void first_v(f32x4& c, FRAG a, FRAG b) {                                  \
      asm volatile(MNEMONIC " %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));                                      \
    }                                                                                                           \
    static __device__ __forceinline__ void acc_a(f32x4& c, FRAG a, FRAG b) {                                    \
      asm volatile(MNEMONIC " %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));                                     \
    }                                                                                                           \
    static __device__ __forceinline__ void acc_v(f32x4& c, FRAG a, FRAG b) {                                    \
      asm volatile(MNEMONIC " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));                                     \
    }                                                                                                           \
  };
RR_MFMA16(_Float16, "v_mfma_f32_16x16x32_f16", f16x8)
RR_MFMA16(__bf16, "v_mfma_f32_16x16x32_bf16", bf16x8)
RR_MFMA16(I8Pair, "v_mfma_i32_16x16x64_i8", s16x8)  // i32 accumulators, converted to f32 (exact: |dot| < 2^24 for 2*D <= 1536) before the epilogue
#undef RR_MFMA16

constexpr int kTicketBatch = 4;  // tiles per dynamic ticket

// LDS ring depth of flat_scan16_kernel.  What must stay in flight per CU is BYTES (~96 KB against the loaded HBM latency:
// 256 CUs x 96 KB / 2.5 us ~ 9.8 TB/s), so narrow rows need more slots; measured with 3 slots: d=384 streamed 3.8 TB/s.
// (INFL-1)*(KG+1) <= 63 (vmcnt range) holds for every entry.
__host__ __device__ constexpr int scan16_slots(int D) { return D >= 640 ? 3 : D == 512 ? 4 : D == 384 ? 6 : D == 256 ? 7 : 13; }

struct LaneState4 {
  float thr[4];
  uint32_t cnt[4], off[4];
};

__device__ __forceinline__ float max4v(const f32x4& v) { return max4(v[0], v[1], v[2], v[3]); }

template <bool DENSE, int NQB, int QPW = 64>
__device__ __forceinline__ void tile_epilogue16(const ScanArgs& a, LaneState4& st, f32x4 (&acc)[2][4], uint32_t j, int lane, int wave) {
  const int col = lane & 15, g = lane >> 4;
  const uint32_t tile = a.tile_first + j * a.tile_stride;
  if (DENSE) {
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      float* d = a.dense + (size_t)(wave * QPW + qb * 16 + col) * a.dense_ld + j * kTileRows + 4 * g;
      *(f32x4*)d = acc[0][qb];
      *(f32x4*)(d + 16) = acc[1][qb];
    }
    return;
  }
  float gm[2][4];
  bool hit = false;
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    gm[0][qb] = max4v(acc[0][qb]);
    gm[1][qb] = max4v(acc[1][qb]);
    hit = hit || (gm[0][qb] > st.thr[qb]) || (gm[1][qb] > st.thr[qb]);
  }
  if (!__builtin_amdgcn_ballot_w64(hit)) return;
  // rare: typically ONE lane with ONE score; only the 4-row group that holds it is expanded
  const uint32_t row0 = tile * kTileRows + 4 * g;
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      if (__builtin_amdgcn_ballot_w64(gm[rb][qb] > st.thr[qb])) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t id = row0 + rb * 16 + i;
          if (acc[rb][qb][i] > st.thr[qb] && id < a.n_rows) {
            a.cand[(size_t)st.off[qb] + st.cnt[qb]] = make_key(acc[rb][qb][i], id);
            ++st.cnt[qb];
          }
        }
      }
    }
  }
  const uint32_t lim = (uint32_t)a.cap - 16u;
  bool full = false;
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) full = full || st.cnt[qb] > lim;
  if (__builtin_amdgcn_ballot_w64(full)) {
    uint64_t* scratch = a.scratch + (size_t)(blockIdx.x * (256 / QPW) + wave) * a.cap;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      uint64_t mask = __builtin_amdgcn_ballot_w64(st.cnt[qb] > lim);
      while (mask) {
        const int L = __builtin_ctzll(mask);
        mask &= mask - 1;
        const uint32_t off = __shfl(st.off[qb], L, 64);
        const int cnt = (int)__shfl(st.cnt[qb], L, 64);
        const uint64_t kth = wave_compact(a.cand + (size_t)__builtin_amdgcn_readfirstlane(off), scratch,
                                          __builtin_amdgcn_readfirstlane(cnt), a.k, lane);
        if (lane == L) { st.cnt[qb] = a.k; st.thr[qb] = key_score(kth); }
      }
    }
  }
}

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
  constexpr int NAQ = NQ < 62 ? NQ : 62;  // ... of which live in AGPRs (4*62 = 248)
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
  const bool norm_wave = L2 && wave == 1;  // not the ticket wave (wave 0)

  // Tile schedule.  Static: ordinals blockIdx, +grid, +2 grid, ...  Dynamic (a.tile_counter): the first two ordinals are
  // static, later ones come in batches of kTicketBatch consecutive ordinals 2*grid + atomicAdd(counter, kTicketBatch)
  // (one dequeue per tile from 256 workgroups would exceed what a single counter word sustains, ~88 per us): XCDs run at different speeds (measured 7.5 % apart
  // under load) and a static split makes every launch wait for the slowest one.  Wave 0 requests the ticket ONE tile
  // ahead with an asm returning atomic that sits in the in-order vmcnt queue before that tile's DMA pieces, so the
  // tile's own vmcnt wait covers it; lane 0 posts it to LDS before the barrier, every wave reads it after.
  const uint32_t stride = gridDim.x;
  const uint32_t n_tiles = a.n_tiles;
  const bool dyn = NS == 3 && a.tile_counter != nullptr;  // (tickets are wired for the 3-slot ring only)
  uint32_t* ticket_lds = (uint32_t*)(smem + NS * TILE_BYTES + NS * 256);  // 2 alternating words
  // The atomic returns asynchronously, so its destination must not be a compiler-visible value (hipcc copies such a
  // register right after the asm statement, before the data lands - seen in the ISA).  It returns into the hard-wired
  // accumulator register a255, claimed through clobbers, and is read inside the same asm statement as the covering vmcnt.
  // A returning atomic is not guaranteed to complete in order with the LDS-DMA loads of the vmcnt queue, so no counted
  // wait can cover it: a255 is pre-loaded with a sentinel and polled at consume time until the return has landed
  // (normally zero spins: the request is a whole tile old).
  const uint32_t one = kTicketBatch, sentinel = 0xFFFFFFFFu;  // a ticket = kTicketBatch consecutive tile ordinals
  auto request_ticket = [&]() {
    if (dyn && wave == 0 && lane == 0)  // ONE lane: every active lane would add to the counter
      // VMEM atomics share one acc bit for data and destination: the addend lives in a254
      asm volatile("v_accvgpr_write_b32 a254, %1\n\tv_accvgpr_write_b32 a255, %2\n\ts_nop 1\n\tglobal_atomic_add a255, %0, a254, off sc0"
                   ::"v"(a.tile_counter), "v"(one), "v"(sentinel) : "memory", "a254", "a255");
  };
  uint32_t j = blockIdx.x, j1 = blockIdx.x + stride;
  request_ticket();
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

  int slot = 0, par = 0, sub = kTicketBatch;
  uint32_t base = 0, dbg_iter = 0;
  while (j < n_tiles) {
    // queue (oldest first): [norms t] DMA t  ticket [norms t+1] DMA t+1 ... -> all but the pieces of the INFL-1 younger tiles
    // (KG each, +1 on the norm wave) are done
    if (norm_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((INFL - 1) * (KG + 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((INFL - 1) * KG) : "memory");
    const bool fetch = dyn && sub == kTicketBatch;  // workgroup-uniform: a new batch of kTicketBatch consecutive tiles starts
    if (fetch && wave == 0) {
      uint32_t ticket;
      do {
        uint32_t tk;
        asm volatile("v_accvgpr_read_b32 %0, a255" : "=v"(tk)::"a255");
        ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
      } while (ticket == sentinel);
      if (lane == 0) ticket_lds[par] = 2 * stride + ticket;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // a raw s_barrier does not wait for the LDS store
    }
    if (a.timeline && threadIdx.x == 0) {  // diagnostics only: tile sequence, and the time the first tile became ready
      if (dbg_iter == 0) a.timeline[3 * gridDim.x + 64 * gridDim.x + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
      a.timeline[3 * gridDim.x + blockIdx.x * 64 + (dbg_iter++ & 63)] = j;
    }
    __builtin_amdgcn_s_barrier();
    int nslot = slot + INFL;
    if (nslot >= NS) nslot -= NS;
    uint32_t j2 = j + INFL * stride;
    if (dyn) {
      if (fetch) {
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket_lds[par]);
        par ^= 1;
        sub = 0;
        request_ticket();  // the next batch: kTicketBatch tiles of lead time, issued before this tile's norms / DMA pieces
      }
      j2 = base + sub;
      ++sub;
    }
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
        const f32x4 h0 = *(const f32x4*)(smem + NS * TILE_BYTES + slot * 256 + (4 * g) * 4);
        const f32x4 h1 = *(const f32x4*)(smem + NS * TILE_BYTES + slot * 256 + (16 + 4 * g) * 4);
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
    j = dyn ? j1 : j + stride;
    j1 = j2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory", "a255");  // no LDS-DMA (or ticket) may outlive the workgroup
  if (!DENSE) {
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) a.cand_cnt[(wave * 64 + qb * 16 + col) * nbuf + blockIdx.x * 4 + g] = st.cnt[qb];
  }
  if (a.timeline && threadIdx.x == 0) {
    a.timeline[gridDim.x + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    a.timeline[2 * gridDim.x + blockIdx.x] = (uint64_t)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  }
}

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
template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }  // body sees a constexpr index
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
template <typename T, bool DENSE, bool L2 = false>
__global__ __launch_bounds__(256, 1) void flat_scan_wide_kernel(const ScanArgs a, const int D) {
  typedef typename Mfma<T>::frag frag;
  constexpr int NT = 8;                    // 32-row tiles per group
  constexpr int STEP_BYTES = NT * 4096;    // one K step of one group in LDS
  constexpr int NS = 3;
  constexpr int LEAD = NS - 1;             // K steps the DMA stream runs ahead
  constexpr int NF = 4 * NT;               // A fragments per K step: (tile, s2, rb)
  constexpr int NB = 8;
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

int g_generic_tall = 16;  // wide-row kernel: 16 = hand-pipelined flat_scan_wide_kernel (default, D % 128 == 0); development A/B: 8 / 4 = compiler-scheduled tall tiles, 0 = the 64-row form
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
static hipError_t launch_scan_generic(const ScanArgs& a, int D, bool dense, int grid, hipStream_t st) {
  if ((g_generic_tall == 16 || a.half_sqnorm) && D % 128 == 0) {
    const size_t lds = 3 * 8 * 4096;
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
  if (a.half_sqnorm) return hipErrorNotSupported;
  if (g_generic_tall == 8) return launch_scan_tall<T, 8>(a, D, dense, grid, st);
  if (g_generic_tall) return launch_scan_tall<T, 4>(a, D, dense, grid, st);
  // two workgroups per CU (<= 256 registers per lane, 16 KB LDS): thread-level parallelism hides the L2 / barrier latency
  if (dense) hipLaunchKernelGGL((flat_scan_generic_kernel<T, true>), dim3(2 * grid), dim3(256), 0, st, a, D);
  else hipLaunchKernelGGL((flat_scan_generic_kernel<T, false>), dim3(2 * grid), dim3(256), 0, st, a, D);
  return hipGetLastError();
}

template <typename T> constexpr bool dtype_is_f16() { return false; }
template <> constexpr bool dtype_is_f16<_Float16>() { return true; }
int g_wide_min_queries = 129;  // 768 < D <= 1536: batches of at least this many queries take the wide-row kernel in one pass (RR_WIDE_MIN_QUERIES)
int g_scan_variant = 16;  // 16 = 16x16x32 pipelined asm (default); dev: 15 same w/o nt, 8 = 8-wave, 1 = 32x32x16 asm, 0 = compiler-scheduled, 3 = generic

static void read_variant_env() {
  static const bool env_read = [] {
    if (const char* v = getenv("RR_SCAN_VARIANT")) g_scan_variant = atoi(v);
    if (const char* v = getenv("RR_GENERIC_TALL")) g_generic_tall = atoi(v);
    if (const char* v = getenv("RR_WIDE_MIN_QUERIES")) g_wide_min_queries = atoi(v);
    return true;
  }();
  (void)env_read;
}

// queries one scan launch serves for this dim: 256, or 128 for the half-resident kernel (768 < D <= 1536)
static bool half_resident_dim(int D) { return D == 896 || D == 1024 || D == 1280 || D == 1536; }
int scan_queries_per_launch(int D, int nq) {
  read_variant_env();
  if (g_scan_variant == 3) return 256;
  return (half_resident_dim(D) && nq < g_wide_min_queries) ? 128 : 256;  // (an L2 search at these dims still works in 128-query blocks: each goes to the wide-row kernel)
}

// candidate buffers per (workgroup, query) of the kernel that will serve this dim
int scan_bufs_per_wg(int D) {
  read_variant_env();
  if (g_scan_variant != 3 && half_resident_dim(D)) return 4;  // half-resident and wide-row kernels: 4 lane quarters
  if (D > kMaxResidentDim || g_scan_variant == 3) return (g_generic_tall == 16 && D % 128 == 0) ? 4 : g_generic_tall ? 2 : 4;  // wide kernel: 4 lane quarters; tall: 2 halves; old: 2 workgroups per CU x 2
  return (g_scan_variant == 0 || g_scan_variant == 1 || g_scan_variant == 2 || (g_scan_variant >= 4 && g_scan_variant <= 7) || g_scan_variant == 9 || g_scan_variant == 48) && D == 768 ? 2 : 4;
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

template <typename T, int D, bool DENSE, bool NT = false, bool L2 = false>
static hipError_t launch_scan16(const ScanArgs& a, int grid, hipStream_t st) {
  const size_t lds = scan16_slots(D) * ((size_t)kTileRows * D * 2 + 256) + 16;  // ring + L2 norm slots + ticket words
  hipError_t e = hipFuncSetAttribute((const void*)flat_scan16_kernel<T, D, DENSE, NT, L2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((flat_scan16_kernel<T, D, DENSE, NT, L2>), dim3(grid), dim3(256), lds, st, a);
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

// Corpora beyond the 256 MiB Infinity Cache are streamed with non-temporal LDS-DMA (read once per batch: -1.7 % at
// B=256, -11 % for single-query searches); smaller ones keep the default policy so back-to-back searches stay on die.
constexpr size_t kNtThresholdBytes = 256ull << 20;

template <typename T, int D>
static hipError_t launch_scan_t(const ScanArgs& a, bool dense, int grid, hipStream_t st) {
  if (D == 768 && g_scan_variant != 16 && !a.half_sqnorm) {  // development variants (A/B, diagnostics) exist for the headline dimension only
    constexpr int DV = 768;
    if (g_scan_variant == 8) return dense ? launch_scan16x8<T, DV, true>(a, grid, st) : launch_scan16x8<T, DV, false>(a, grid, st);
    if (g_scan_variant == 0) return dense ? launch_scan_v<T, DV, true, 0>(a, grid, st) : launch_scan_v<T, DV, false, 0>(a, grid, st);
    if (g_scan_variant == 1) return dense ? launch_scan_v<T, DV, true, 1>(a, grid, st) : launch_scan_v<T, DV, false, 1>(a, grid, st);
    if (g_scan_variant == 2 && !dense) return launch_scan_v<T, DV, false, 2>(a, grid, st);
    if (g_scan_variant == 2) return launch_scan_v<T, DV, true, 1>(a, grid, st);
    if (g_scan_variant == 15 && !dense) return launch_scan16<T, DV, false, false>(a, grid, st);  // 16x16x32 without nt
#ifdef RR_ABLATION_VARIANTS
    if (!dense) {
      if (g_scan_variant == 4) return launch_scan_v<T, DV, false, 4>(a, grid, st);
      if (g_scan_variant == 5) return launch_scan_v<T, DV, false, 5>(a, grid, st);
      if (g_scan_variant == 6) return launch_scan_v<T, DV, false, 6>(a, grid, st);
      if (g_scan_variant == 7) return launch_scan_v<T, DV, false, 7>(a, grid, st);
      if (g_scan_variant == 48) return launch_scan_v<T, DV, false, 48>(a, grid, st);
      if (g_scan_variant == 9 && dtype_is_f16<T>()) return launch_scan_v<T, DV, false, 9>(a, grid, st);
    }
    if (g_scan_variant >= 4) return launch_scan_v<T, DV, true, 1>(a, grid, st);
#endif
  }
  const bool nt = (size_t)a.n_rows * D * 2 > kNtThresholdBytes;
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
  if (a.half_sqnorm && g_scan_variant == 3) return hipErrorNotSupported;  // L2: the resident-query and wide-row kernels only
  if (g_scan_variant != 3 && half_resident_dim(D) && (((int)a.nq >= g_wide_min_queries && g_generic_tall == 16) || a.half_sqnorm)) {
    // more than 128 queries: one pass of the wide-row kernel beats two passes of the half-resident one
    if (dtype == RR_DTYPE_F16) return launch_scan_generic<_Float16>(a, D, dense, grid, st);
    if (dtype == RR_DTYPE_BF16) return launch_scan_generic<__bf16>(a, D, dense, grid, st);
    return hipErrorInvalidValue;
  }
  if (g_scan_variant != 3 && half_resident_dim(D)) {  // 768 < D <= 1536: 32 resident queries per wave, half-tile ring
    if (dtype == RR_DTYPE_F16) return launch_scan_half_resident<_Float16>(a, D, dense, grid, st);
    if (dtype == RR_DTYPE_BF16) return launch_scan_half_resident<__bf16>(a, D, dense, grid, st);
    return hipErrorInvalidValue;
  }
  if (D > kMaxResidentDim || g_scan_variant == 3) {  // generic-dimension kernel (also forced by RR_SCAN_VARIANT=3)
    if (D % 64 != 0) return hipErrorInvalidValue;
    if (dtype == RR_DTYPE_F16) return launch_scan_generic<_Float16>(a, D, dense, grid, st);
    if (dtype == RR_DTYPE_BF16) return launch_scan_generic<__bf16>(a, D, dense, grid, st);
    return hipErrorInvalidValue;
  }
  if (dtype == RR_DTYPE_F16) return launch_scan_d<_Float16>(a, D, dense, grid, st);
  if (dtype == RR_DTYPE_BF16) return launch_scan_d<__bf16>(a, D, dense, grid, st);
  return hipErrorInvalidValue;
}

int scan_padded_dim(int d) {
  static const int dims[] = {128, 256, 384, 512, 640, 768, 896, 1024, 1280, 1536};  // query-resident instantiations
  read_variant_env();
  for (int v : dims)
    if (d <= v && (v <= kMaxResidentDim || g_scan_variant != 3)) return v;
  if (d <= kMaxDim) return (d + 127) / 128 * 128;  // wide-row kernel
  return -1;
}

}  // namespace rr
