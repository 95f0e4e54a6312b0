// K2 / K4: exact selection kernels around the scan.
//  * dense_select : top-k (or just the k-th score, for the bootstrap threshold) of a dense score row
//  * compact      : fold every workgroup's candidate buffers into the running per-query top-k and
//                   publish the next (strict) threshold
//  * finalize     : running list -> faiss-shaped (D f32[nq,k], I i64[nq,k])
//  * merge_topk   : cross-source (score, id) merge — reference ragroute/rerank.py:3-9, 28-34
// All of them order candidates by (score descending, id ascending); one workgroup per query,
// bitonic sort of 64-bit order keys in LDS.
#include "rr_common.h"
#include "rr_kernels.h"

namespace rr {

__device__ __forceinline__ float next_below(float x) {
  // largest float strictly below x (x finite or -inf); s > next_below(t)  <=>  s >= t
  if (x == -__builtin_inff()) return x;
  uint32_t b = __float_as_uint(x);
  if ((b & 0x7FFFFFFFu) == 0) return __uint_as_float(0x80000001u);
  return __uint_as_float((b & 0x80000000u) ? b + 1 : b - 1);
}

__device__ __forceinline__ int pow2_ceil(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

// descending bitonic sort of n (power of two) u64 keys in LDS by the whole workgroup
__device__ void bitonic_sort_desc(uint64_t* s, int n) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int k2 = 2; k2 <= n; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n; i += nt) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t x = s[i], y = s[ixj];
          const bool desc = (i & k2) == 0;
          if ((x < y) == desc) { s[i] = y; s[ixj] = x; }
        }
      }
      __syncthreads();
    }
  }
}

__global__ void init_state_kernel(SelectArgs a) {
  const uint32_t q = threadIdx.x;
  if (q < kQueriesPerBlock) {
    a.list_cnt[q] = 0;
    a.thr[q] = q < a.nq ? -__builtin_inff() : __builtin_inff();
  }
}

template <bool BOOTSTRAP>
__global__ __launch_bounds__(1024) void dense_select_kernel(SelectArgs a) {
  __shared__ uint64_t keys[kSelectCap];
  const uint32_t q = blockIdx.x;
  if (q >= a.nq) return;
  const int n = (int)a.dense_cols;
  const int np = pow2_ceil(n < 2 ? 2 : n);
  for (int c = threadIdx.x; c < np; c += blockDim.x) {
    uint64_t key = 0;
    if (c < n) {
      const uint32_t row = (a.tile_first + (uint32_t)(c >> 5) * a.tile_stride) * kTileRows + (c & 31);
      const float s = a.dense[(size_t)q * a.dense_ld + c];
      if (row < a.n_rows && s == s) key = make_key(s, row);
    }
    keys[c] = key;
  }
  __syncthreads();
  bitonic_sort_desc(keys, np);
  if (BOOTSTRAP) {
    if (threadIdx.x == 0 && a.k <= np) {
      const uint64_t kth = keys[a.k - 1];
      if ((kth >> 32) != 0) a.thr[q] = next_below(key_score(kth));
    }
  } else {
    for (int i = threadIdx.x; i < a.k; i += blockDim.x) {
      const uint64_t key = i < np ? keys[i] : 0;
      a.list[(size_t)q * a.list_ld + i] = key;
      // the first empty slot (or k) is the count
      const bool valid = (key >> 32) != 0;
      const bool next_valid = (i + 1 < a.k) && (i + 1 < np) && ((keys[i + 1] >> 32) != 0);
      if (valid && !next_valid) a.list_cnt[q] = i + 1;
    }
  }
}

__global__ __launch_bounds__(1024) void compact_kernel(SelectArgs a) {
  __shared__ uint64_t keys[kSelectCap];
  __shared__ int scan[1024];
  const uint32_t q = blockIdx.x;
  if (q >= a.nq) return;
  const int tid = threadIdx.x;
  const int k = a.k, cap = a.cap;
  int fill = (int)a.list_cnt[q];
  for (int i = tid; i < fill; i += blockDim.x) keys[i] = a.list[(size_t)q * a.list_ld + i];
  int G = (kSelectCap - kMaxK) / cap;  // buffers appended per round; a round always fits after a truncation
  if (G > 1024) G = 1024;
  auto sort_truncate = [&]() {
    const int np = pow2_ceil(fill < 2 ? 2 : fill);
    for (int i = fill + tid; i < np; i += blockDim.x) keys[i] = 0;
    __syncthreads();
    bitonic_sort_desc(keys, np);
    if (fill > k) fill = k;
  };
  for (uint32_t g0 = 0; g0 < a.nbuf; g0 += G) {
    const uint32_t b = g0 + tid;
    int c = 0;
    if (tid < G && b < a.nbuf) {
      c = (int)a.cand_cnt[q * a.nbuf + b];
      if (c > cap) c = cap;
    }
    scan[tid] = c;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // inclusive Hillis-Steele scan
      const int v = tid >= d ? scan[tid - d] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    const int tot = scan[1023];
    const int offs = scan[tid] - c;
    if (tot == 0) { __syncthreads(); continue; }
    if (fill + tot > kSelectCap) sort_truncate();  // uniform: fill and tot are workgroup-uniform
    if (c > 0) {
      const uint64_t* src = a.cand + ((size_t)q * a.nbuf + b) * cap;
      for (int e = 0; e < c; ++e) keys[fill + offs + e] = src[e];
    }
    fill += tot;
    __syncthreads();
  }
  sort_truncate();
  for (int i = tid; i < k; i += blockDim.x) a.list[(size_t)q * a.list_ld + i] = i < fill ? keys[i] : 0;
  if (tid == 0) {
    a.list_cnt[q] = fill;
    // strict threshold: every later row has a larger id than the k listed rows, so a tie loses
    if (fill == k) a.thr[q] = key_score(keys[k - 1]);
  }
}

__global__ void finalize_kernel(SelectArgs a, float* D, int64_t* I, int64_t id_offset) {
  const uint32_t q = blockIdx.x;
  if (q >= a.nq) return;
  const int cnt = (int)a.list_cnt[q];
  for (int i = threadIdx.x; i < a.k; i += blockDim.x) {
    float s = -__builtin_inff();
    int64_t id = -1;
    if (i < cnt) {
      const uint64_t key = a.list[(size_t)q * a.list_ld + i];
      s = key_score(key);
      id = (int64_t)key_id(key) + id_offset;
    }
    D[(size_t)q * a.k + i] = s;
    I[(size_t)q * a.k + i] = id;
  }
}

// ---- cross-source merge (K4) ---------------------------------------------------------------
// (ord', id) pairs with 64-bit ids: ord' = ord(score) for descending, ~ord(score) for ascending,
// 0 for padding / NaN.  better(a,b) = a.s > b.s || (a.s == b.s && a.id < b.id).
__global__ __launch_bounds__(1024) void merge_topk_kernel(const float* Din, const int64_t* Iin, int m, int k,
                                                          int descending, float* Dout, int64_t* Iout) {
  __shared__ uint32_t ss[kSelectCap];
  __shared__ int64_t ids[kSelectCap];
  const uint32_t q = blockIdx.x;
  const int np = pow2_ceil(m < 2 ? 2 : m);
  for (int c = threadIdx.x; c < np; c += blockDim.x) {
    uint32_t s = 0;
    int64_t id = INT64_MAX;
    if (c < m) {
      const float v = Din[(size_t)q * m + c];
      const int64_t i = Iin[(size_t)q * m + c];
      if (i >= 0 && v == v) {
        s = descending ? ord_f32(v) : ~ord_f32(v);
        id = i;
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
          const int64_t ix = ids[i], iy = ids[ixj];
          const bool x_worse = sx < sy || (sx == sy && ix > iy);
          const bool desc = (i & k2) == 0;
          if (x_worse == desc && !(sx == sy && ix == iy)) {
            ss[i] = sy; ss[ixj] = sx; ids[i] = iy; ids[ixj] = ix;
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
    }
    Dout[(size_t)q * k + i] = v;
    Iout[(size_t)q * k + i] = id;
  }
}

hipError_t launch_init_state(const SelectArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_dense_select(const SelectArgs& a, bool bootstrap, hipStream_t st) {
  if (bootstrap) hipLaunchKernelGGL(dense_select_kernel<true>, dim3(kQueriesPerBlock), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL(dense_select_kernel<false>, dim3(kQueriesPerBlock), dim3(1024), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_compact(const SelectArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(compact_kernel, dim3(kQueriesPerBlock), dim3(1024), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_finalize(const SelectArgs& a, float* D, int64_t* I, int64_t id_offset, hipStream_t st) {
  hipLaunchKernelGGL(finalize_kernel, dim3(kQueriesPerBlock), dim3(256), 0, st, a, D, I, id_offset);
  return hipGetLastError();
}
hipError_t launch_merge_topk(const float* Din, const int64_t* Iin, int nq, int m, int k, int descending, float* Dout,
                             int64_t* Iout, hipStream_t st) {
  hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(1024), 0, st, Din, Iin, m, k, descending, Dout, Iout);
  return hipGetLastError();
}

}  // namespace rr
