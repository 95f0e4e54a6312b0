// Device helpers shared by the scan kernels (flat_scan.hip: production kernels; flat_scan_dev.hip: measured alternatives
// kept for A/B runs): MFMA wrappers, asm LDS / wait primitives, the exact wave-level compaction and the tile epilogues.
#pragma once
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
static __device__ __forceinline__ uint64_t wave_compact_inl(uint64_t* buf, uint64_t* scratch, int cnt, int k, int lane) {
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
static __device__ __noinline__ uint64_t wave_compact(uint64_t* buf, uint64_t* scratch, int cnt, int k, int lane) {
  return wave_compact_inl(buf, scratch, cnt, k, lane);
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

// LDS ring depth of flat_scan16_kernel.  What must stay in flight per CU is BYTES (~96 KB against the loaded HBM latency:
// 256 CUs x 96 KB / 2.5 us ~ 9.8 TB/s), so narrow rows need more slots; measured with 3 slots: d=384 streamed 3.8 TB/s.
// (INFL-1)*(KG+1) <= 63 (vmcnt range) holds for every entry.
__host__ __device__ constexpr int scan16_slots(int D) { return D >= 640 ? 3 : D == 512 ? 4 : D == 384 ? 6 : D == 256 ? 7 : 13; }

// LDS reads hidden from hipcc: next to LDS-DMA it answers every LDS load it can see with s_waitcnt vmcnt(0), which drains
// the DMA ring of the scan kernels.  Each statement waits for its own data (the fragment ring is empty wherever these are used).
__device__ __forceinline__ float lds_load_f32(uint32_t byte_addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr) : "memory");
  return v;
}
__device__ __forceinline__ f32x4 lds_load_f32x4(uint32_t byte_addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr) : "memory");
  return v;
}

// Loads of the rare paths, with their own wait and invisible to hipcc's waitcnt pass: a VMEM load it can see inside a loop that
// keeps LDS-DMA / prefetched registers in flight makes it put s_waitcnt vmcnt(0) into the hot path (pitfall i).
// The wait drains the wave's DMA ring too, which is always safe (the counted waits that follow are then met at once).
__device__ __forceinline__ uint32_t asm_load_u32(const void* p) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ float asm_load_f32(const void* p) { return __uint_as_float(asm_load_u32(p)); }

// ---- segmented search: launch ordinal -> tile --------------------------------------------------------------------------------
// A chunk launch of a segmented search covers one run of tiles per segment (ScanArgs::ranges).  A workgroup's ordinals only move
// forward, so each stream of a kernel (DMA issue, query prefetch, epilogue) keeps a wave-uniform cursor: the run it is in, where
// the run ends and the offset to add.  Plain launches: j_end = ~0 (never reached), delta = 0, and the tile is
// tile_first + j * tile_stride as before (segmented chunk launches have tile_first = 0, tile_stride = 1).
struct TileCursor {
  uint32_t j_end;
  int32_t delta;
  int r;
};
__device__ __forceinline__ void cursor_init(const ScanArgs& a, TileCursor& c) {
  c.r = -1;
  c.delta = 0;
  c.j_end = a.ranges ? 0u : 0xFFFFFFFFu;
}
// (rare: at most n_ranges times per launch and stream) move to the run holding ordinal j; returns its RangeEntry.
// The entry comes through the scalar cache (s_load + lgkmcnt(0): the callers sit where the wave's LDS fragment ring is empty), so
// the DMA ring is NOT drained here.
__device__ __forceinline__ RangeEntry cursor_advance(const ScanArgs& a, TileCursor& c, uint32_t j) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  RangeEntry e;
  do {
    ++c.r;
    u32x4 v;
    const RangeEntry* p = a.ranges + c.r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    e.j_end = v[0];
    e.delta = (int32_t)v[1];
    e.seg = v[2];
    e.row_limit = v[3];
  } while (j >= e.j_end);   // (the last run's j_end is ~0: ordinals past the launch's end - streams run ahead - stay in it)
  c.j_end = e.j_end;
  c.delta = e.delta;
  return e;
}
// the lane's thresholds of segment `seg` for up to four of its queries: all loads in flight together, ONE wait (it drains the
// wave's DMA ring: once per segment slice and launch)
template <int N>
__device__ __forceinline__ void load_thresholds(const ScanArgs& a, uint32_t seg, const uint32_t (&qi)[4], float (&thr)[4]) {
  const float* base = a.thr + (size_t)seg * kQueriesPerBlock;
  const float *p0 = base + qi[0], *p1 = base + qi[N > 1 ? 1 : 0], *p2 = base + qi[N > 2 ? 2 : 0], *p3 = base + qi[N > 3 ? 3 : 0];
  float t0, t1, t2, t3;
  asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %5, off\n\tglobal_load_dword %2, %6, off\n\t"
               "global_load_dword %3, %7, off\n\ts_waitcnt vmcnt(0)"
               : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
  thr[0] = t0;
  if (N > 1) thr[1] = t1;
  if (N > 2) thr[2] = t2;
  if (N > 3) thr[3] = t3;
}
__device__ __forceinline__ uint32_t cursor_tile(const ScanArgs& a, TileCursor& c, uint32_t j) {
  if (j >= c.j_end) (void)cursor_advance(a, c, j);
  return a.tile_first + j * a.tile_stride + (uint32_t)c.delta;
}

struct LaneState4 {
  float thr[4];
  uint32_t cnt[4], off[4];
  // segmented search (wave-uniform): the epilogue's cursor, and the first row past the valid rows of the segment it is in.
  // Plain search: row_limit = n_rows.
  TileCursor cur;
  uint32_t row_limit;
};

__device__ __forceinline__ void lane_state_segments_init(const ScanArgs& a, LaneState4& st) {
  cursor_init(a, st.cur);
  st.row_limit = a.n_rows;
}

__device__ __forceinline__ float max4v(const f32x4& v) { return max4(v[0], v[1], v[2], v[3]); }

// INLINE_COMPACT: no function call in the slow path (for kernels that keep asynchronously loaded registers live across it)
// QPW: queries per wave (the wave's queries are wave * QPW + qb * 16 + col); WAVES: waves per workgroup (compaction scratch slots);
// DEAL: the query blocks are dealt round robin instead (slot qb of wave w = block 4 qb + w: flat_scan16_kernel)
// qwave: the wave index the lane's QUERIES are derived from, where that is not the wave itself (row-split kernel: two waves share a
// query quarter; `wave` still names the wave's own compaction scratch)
// SEG: the launch may be a chunk of a segmented search (ScanArgs::ranges / ties_pass are honoured); false compiles the plain
// search's arithmetic only (no cursor compare per tile, no ties_pass select per refresh: flat_scan16_kernel's plain instantiation)
template <bool DENSE, int NQB, int QPW = 64, bool INLINE_COMPACT = false, int WAVES = 256 / QPW, bool DEAL = false, bool SEG = true>
__device__ __forceinline__ void tile_epilogue16(const ScanArgs& a, LaneState4& st, f32x4 (&acc)[2][4], uint32_t j, int lane, int wave,
                                                int qwave = -1) {
  const int col = lane & 15, g = lane >> 4;
  const int qw = qwave < 0 ? wave : qwave;
  if (DENSE) {
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      float* d = a.dense + (size_t)(DEAL ? (qb * 4 + qw) * 16 + col : qw * QPW + qb * 16 + col) * a.dense_ld + j * kTileRows + 4 * g;
      *(f32x4*)d = acc[0][qb];
      *(f32x4*)(d + 16) = acc[1][qb];
    }
    return;
  }
  if (SEG && j >= st.cur.j_end) {   // segmented search only (plain: j_end = ~0); wave-uniform, once per segment and launch: the wave enters
    // another segment's slice - its valid-row limit and its thresholds (+inf for queries not routed to it) replace the lane's
    const RangeEntry e = cursor_advance(a, st.cur, j);
    st.row_limit = e.row_limit;
    uint32_t qi[4];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) qi[qb] = DEAL ? (qb * 4 + qw) * 16 + col : qw * QPW + (qb < NQB ? qb : 0) * 16 + col;
    load_thresholds<NQB>(a, e.seg, qi, st.thr);
  }
  const uint32_t tile = a.tile_first + j * a.tile_stride + (SEG ? (uint32_t)st.cur.delta : 0u);
#ifndef RR_EPILOGUE_MASKS
#define RR_EPILOGUE_MASKS 0   // 1: the insertion path branches on wave masks made by the filter instead of re-deriving them per group (measured: headline -0.4 %, config 2 +0.8 %, i.e. noise; off)
#endif
#if RR_EPILOGUE_MASKS
  uint64_t hm[2][4];          // lanes whose 4-row group of (row block, query block) holds a score above the lane's threshold
  uint64_t any = 0;
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    hm[0][qb] = __builtin_amdgcn_ballot_w64(max4v(acc[0][qb]) > st.thr[qb]);
    hm[1][qb] = __builtin_amdgcn_ballot_w64(max4v(acc[1][qb]) > st.thr[qb]);
    any |= hm[0][qb] | hm[1][qb];
  }
  if (!any) return;
#else
  float gm[2][4];
  bool hit = false;
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    gm[0][qb] = max4v(acc[0][qb]);
    gm[1][qb] = max4v(acc[1][qb]);
    hit = hit || (gm[0][qb] > st.thr[qb]) || (gm[1][qb] > st.thr[qb]);
  }
  if (!__builtin_amdgcn_ballot_w64(hit)) return;
#endif
  // rare: typically ONE lane with ONE score; only the 4-row group that holds it is expanded
  const uint32_t row0 = tile * kTileRows + 4 * g;
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#if RR_EPILOGUE_MASKS
      if (hm[rb][qb]) {
#else
      if (__builtin_amdgcn_ballot_w64(gm[rb][qb] > st.thr[qb])) {
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t id = row0 + rb * 16 + i;
          if (acc[rb][qb][i] > st.thr[qb] && id < st.row_limit) {
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
    uint64_t* scratch = a.scratch + (size_t)(blockIdx.x * WAVES + wave) * a.cap;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      uint64_t mask = __builtin_amdgcn_ballot_w64(st.cnt[qb] > lim);
      while (mask) {
        const int L = __builtin_ctzll(mask);
        mask &= mask - 1;
        const uint32_t off = __shfl(st.off[qb], L, 64);
        const int cnt = (int)__shfl(st.cnt[qb], L, 64);
        uint64_t* buf = a.cand + (size_t)__builtin_amdgcn_readfirstlane(off);
        const uint64_t kth = INLINE_COMPACT ? wave_compact_inl(buf, scratch, __builtin_amdgcn_readfirstlane(cnt), a.k, lane)
                                            : wave_compact(buf, scratch, __builtin_amdgcn_readfirstlane(cnt), a.k, lane);
        if (lane == L) { st.cnt[qb] = a.k; st.thr[qb] = (SEG && a.ties_pass) ? next_below(key_score(kth)) : key_score(kth); }
      }
    }
    // Inlined, the compaction's own global loads are visible to hipcc's waitcnt pass: without a wait IT can see, it treats
    // them as possibly pending around the caller's loop and answers with s_waitcnt vmcnt(0) inside the hot step (which
    // drains the LDS-DMA ring on every K step).  vmcnt(0), expcnt / lgkmcnt untouched:
    if (INLINE_COMPACT) __builtin_amdgcn_s_waitcnt(0x0F70);
  }
}

template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }  // body sees a constexpr index

// Corpora beyond the 256 MiB Infinity Cache are streamed with non-temporal LDS-DMA (read once per batch: -1.7 % at
// B=256, -11 % for single-query searches); smaller ones keep the default policy so back-to-back searches stay on die.
constexpr size_t kNtThresholdBytes = 256ull << 20;

// flat_scan_wide.hip
#ifndef RR_WIDE_QFRAG
#define RR_WIDE_QFRAG 1   // the pd / wide8 kernels' query loads read prep_kernel's fragment-order copy (one contiguous KiB per
#endif                    // instruction) instead of 16 rows x 64 B of the row-major block; 0 = row-major (A/B builds)
hipError_t launch_scan_wide(const ScanArgs& a, int dtype, int D, bool dense, int grid, hipStream_t st);
const char* scan_wide_kernel_name(int D, int nq);
bool scan_wide_rowsplit(int D, int nq, bool l2, int k);   // the row-split kernel serves this launch (8 candidate buffers per workgroup and query)

// flat_scan_dev.hip (only in builds with -DRR_DEV_VARIANTS): the development kernels behind RR_SCAN_VARIANT / RR_GENERIC_TALL
bool dev_scan_handles(const ScanArgs& a, int D, int variant, int tall);
int dev_scan_bufs_per_wg(int D, int variant, int tall);
hipError_t launch_dev_scan(const ScanArgs& a, int dtype, int D, bool dense, int grid, hipStream_t st, int variant, int tall);

}  // namespace rr
