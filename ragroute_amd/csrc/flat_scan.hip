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
//  * Per tile: KS x { ds_read_b128 corpus fragment ; 2 x v_mfma_f32_32x32x16 } with the corpus as
//    A and the queries as B, so every lane ends up owning ONE query (column) and 16 corpus rows:
//    the top-k filter is a per-lane compare against that query's threshold, no cross-lane work.
//  * Scores strictly above the threshold are appended (as 64-bit order keys) to a private
//    per-(workgroup, query, lane-half) buffer; if a buffer fills, the wave compacts it exactly
//    to its k best and raises that lane's threshold.  Nothing is ever dropped that could be in
//    the final top-k (see DESIGN.md "exactness").
//  * DENSE=true writes every score instead (bootstrap sample and tiny corpora).
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

template <typename T, int D, bool DENSE>
__global__ __launch_bounds__(256, 1) void flat_scan_kernel(const ScanArgs a) {
  typedef typename Mfma<T>::frag frag;
  constexpr int KS = D / 16;  // 16-wide k slices (one MFMA each per query block)
  constexpr int KG = D / 64;  // 64-wide k groups (one 1 KiB DMA piece per 8 rows)
  constexpr int TILE_BYTES = kTileRows * D * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const uint32_t q0i = wave * 64 + r, q1i = q0i + 32;

  // ---- resident queries (MFMA B operand: lane holds query r, k = 16 s + 8 h .. +7) ----------
  frag q0[KS], q1[KS];
  {
    const T* xq = (const T*)a.xq;
    const T* p0 = xq + (size_t)q0i * D + 8 * h;
    const T* p1 = xq + (size_t)q1i * D + 8 * h;
    const bool v0 = q0i < a.nq, v1 = q1i < a.nq;
    const frag z = {0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      q0[s] = v0 ? *(const frag*)(p0 + 16 * s) : z;
      q1[s] = v1 ? *(const frag*)(p1 + 16 * s) : z;
    }
  }

  float thr0 = 0.f, thr1 = 0.f;
  uint32_t cnt0 = 0, cnt1 = 0, off0 = 0, off1 = 0;
  const uint32_t nbuf = gridDim.x * 2;
  if (!DENSE) {
    thr0 = a.thr[q0i];
    thr1 = a.thr[q1i];
    off0 = (q0i * nbuf + blockIdx.x * 2 + h) * (uint32_t)a.cap;
    off1 = (q1i * nbuf + blockIdx.x * 2 + h) * (uint32_t)a.cap;
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

  auto issue_tile = [&](uint32_t j, int slot) {
    const uint32_t tile = a.tile_first + j * a.tile_stride;
    uint32_t row = tile * kTileRows + wave * 8 + rho_w;
    row = row < a.n_rows ? row : a.n_rows - 1;
    const char* g = (const char*)a.xb + (size_t)row * (D * 2) + c_w * 16;
    char* l = smem + slot * TILE_BYTES + wave * 1024;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + kg * 128),
                                       (__attribute__((address_space(3))) void*)(l + kg * 4096), 16, 0, 0);
  };

  uint32_t j = blockIdx.x;
  const uint32_t stride = gridDim.x;
  const uint32_t n_tiles = a.n_tiles;
  if (j < n_tiles) issue_tile(j, 0);
  if (j + stride < n_tiles) issue_tile(j + stride, 1);
  int slot = 0;
  for (; j < n_tiles; j += stride) {
    if (j + stride < n_tiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KG) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int nslot = slot + 2;
    if (nslot >= 3) nslot -= 3;
    if (j + 2 * stride < n_tiles) issue_tile(j + 2 * stride, nslot);

    f32x16 a0 = {0}, a1 = {0};
    const char* base = smem + slot * TILE_BYTES;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const frag c = *(const frag*)(base + roff[s & 3] + (s >> 2) * 4096);
      a0 = Mfma<T>::run(c, q0[s], a0);
      a1 = Mfma<T>::run(c, q1[s], a1);
    }

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
      float m0 = a0[0], m1 = a1[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) {
        m0 = fmaxf(m0, a0[i]);
        m1 = fmaxf(m1, a1[i]);
      }
      if (__builtin_amdgcn_ballot_w64(m0 > thr0 || m1 > thr1)) {
        const uint32_t row0 = tile * kTileRows + 4 * h;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const uint32_t id = row0 + (i & 3) + 8 * (i >> 2);
          if (a0[i] > thr0 && id < a.n_rows) {
            a.cand[(size_t)off0 + cnt0] = make_key(a0[i], id);
            ++cnt0;
          }
          if (a1[i] > thr1 && id < a.n_rows) {
            a.cand[(size_t)off1 + cnt1] = make_key(a1[i], id);
            ++cnt1;
          }
        }
        // keep >= 16 free slots per buffer; compaction is exact and raises the lane threshold
        const uint32_t lim = (uint32_t)a.cap - 16u;
        if (__builtin_amdgcn_ballot_w64(cnt0 > lim || cnt1 > lim)) {
          uint64_t* scratch = a.scratch + (size_t)(blockIdx.x * 4 + wave) * a.cap;
#pragma unroll 1
          for (int b = 0; b < 2; ++b) {
            uint64_t mask = __builtin_amdgcn_ballot_w64((b ? cnt1 : cnt0) > lim);
            while (mask) {
              const int L = __builtin_ctzll(mask);
              mask &= mask - 1;
              const uint32_t off = __shfl(b ? off1 : off0, L, 64);
              const int cnt = (int)__shfl(b ? cnt1 : cnt0, L, 64);
              const uint64_t kth = wave_compact(a.cand + (size_t)__builtin_amdgcn_readfirstlane(off), scratch,
                                                __builtin_amdgcn_readfirstlane(cnt), a.k, lane);
              if (lane == L) {
                if (b) { cnt1 = a.k; thr1 = key_score(kth); }
                else   { cnt0 = a.k; thr0 = key_score(kth); }
              }
            }
          }
        }
      }
    }
    slot = slot + 1;
    if (slot >= 3) slot = 0;
  }
  if (!DENSE) {
    a.cand_cnt[q0i * nbuf + blockIdx.x * 2 + h] = cnt0;
    a.cand_cnt[q1i * nbuf + blockIdx.x * 2 + h] = cnt1;
  }
}

template <typename T, int D>
static hipError_t launch_scan_t(const ScanArgs& a, bool dense, int grid, hipStream_t st) {
  const size_t lds = 3 * (size_t)kTileRows * D * 2;
  hipError_t e;
  if (dense) {
    e = hipFuncSetAttribute((const void*)flat_scan_kernel<T, D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((flat_scan_kernel<T, D, true>), dim3(grid), dim3(256), lds, st, a);
  } else {
    e = hipFuncSetAttribute((const void*)flat_scan_kernel<T, D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((flat_scan_kernel<T, D, false>), dim3(grid), dim3(256), lds, st, a);
  }
  return hipGetLastError();
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

hipError_t launch_flat_scan(const ScanArgs& a, int dtype, int D, bool dense, int grid, hipStream_t st) {
  if (dtype == RR_DTYPE_F16) return launch_scan_d<_Float16>(a, D, dense, grid, st);
  if (dtype == RR_DTYPE_BF16) return launch_scan_d<__bf16>(a, D, dense, grid, st);
  return hipErrorInvalidValue;
}

int scan_padded_dim(int d) {
  static const int dims[] = {128, 256, 384, 512, 640, 768};
  for (int v : dims)
    if (d <= v) return v;
  return -1;
}

}  // namespace rr
