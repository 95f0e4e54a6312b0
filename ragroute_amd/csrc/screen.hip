// Exact search through an int8 screening copy (rr_flat_search_screened).
//
// The corpus keeps its f16/bf16 rows and, beside them, one int8 copy x8 = round(x / s) with ONE scale s for the whole
// corpus.  A search quantises the queries the same way (one scale per query), streams the int8 copy (half the HBM bytes,
// 2x the MFMA rate) through the same scan kernel and the same exact top-k machinery with k' = list_len >> k, re-scores
// those list_len rows per query from the f16/bf16 rows, and proves the result:
//     S(q,x) = q.x = sq*s*P + sq*q8.e_x + e_q.x       (P = q8.x8 integer dot, e_* = quantisation residuals)
//  => |S - sq*s*P| <= |sq q8| * max_x|e_x| + |e_q| * max_x|x| =: eps(q)
// every row outside the list has P <= P_L (the list's smallest P), hence S <= sq*s*P_L + eps; if that is strictly below
// the k-th best re-scored value, the k best re-scored rows ARE the exact top-k (d_exact[q] = 1).  Otherwise d_exact[q] = 0
// and the caller runs rr_flat_search for that batch (ragroute_amd.flat_index does).  No reference counterpart: the
// reference's faiss.IndexFlatIP always scans f32 rows (data_source.py:158,186,203); this keeps its results, not its bytes.
#include "rr_common.h"
#include "rr_kernels.h"
#include "rr_sort.h"

namespace rr {

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// stats words: [0] max |x| over the corpus, [1] max over rows of |x - s*x8|^2, [2] max over rows of |x|^2 (f32 bit patterns of
// non-negative values: unsigned integer max == float max)
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* __restrict__ xb, int64_t n_chunks, uint32_t* stats) {
  typedef T vec8 __attribute__((ext_vector_type(8)));
  float m = 0.f;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * blockDim.x) {
    const vec8 v = *(const vec8*)(xb + c * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf((float)v[i]));
  }
  m = wmax(m);
  if ((threadIdx.x & 63) == 0) atomicMax(&stats[0], __float_as_uint(m));
}

__device__ __forceinline__ float scale_of(float amax) { return amax > 0.f ? amax * (1.0f / 127.0f) : 1.0f; }

// 8 values -> 8 int8 (round to nearest, clamped to +-127) packed in a uint2; accumulates the residual and the norm
__device__ __forceinline__ uint2 quantise8(const float (&x)[8], float s, float inv, float& e2, float& n2) {
  uint32_t w[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float r = rintf(x[i] * inv);
    r = fminf(fmaxf(r, -127.f), 127.f);
    const float err = fmaf(-s, r, x[i]);
    e2 = fmaf(err, err, e2);
    n2 = fmaf(x[i], x[i], n2);
    w[i >> 2] |= ((uint32_t)(int)r & 0xFFu) << (8 * (i & 3));
  }
  return make_uint2(w[0], w[1]);
}

// one wave per row, rows grid-strided; int8 row width dim8 >= dim (zero padded)
template <typename T>
__global__ __launch_bounds__(256) void quantise_rows_kernel(const T* __restrict__ xb, int64_t n, int dim, int8_t* __restrict__ x8,
                                                            int dim8, uint32_t* stats) {
  typedef T vec8 __attribute__((ext_vector_type(8)));
  const int lane = threadIdx.x & 63;
  const int64_t wpb = blockDim.x >> 6;
  const float s = scale_of(__uint_as_float(stats[0])), inv = 1.0f / s;
  float e2max = 0.f, n2max = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * wpb) {
    float e2 = 0.f, n2 = 0.f;
    for (int c = lane; c < dim8 / 8; c += 64) {
      float x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < dim / 8) {
        const vec8 v = *(const vec8*)(xb + row * dim + c * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (float)v[i];
      }
      *(uint2*)(x8 + row * dim8 + c * 8) = quantise8(x, s, inv, e2, n2);
    }
    e2max = fmaxf(e2max, wsum(e2));
    n2max = fmaxf(n2max, wsum(n2));
  }
  if (lane == 0) {
    atomicMax(&stats[1], __float_as_uint(e2max));
    atomicMax(&stats[2], __float_as_uint(n2max));
  }
}

// one wave per query: own scale, int8 row, and qinfo[q] = {sq*s (score per integer unit), eps(q)}
template <typename T>
__global__ __launch_bounds__(256) void quantise_queries_kernel(const T* __restrict__ xq, int nq, int dim, int8_t* __restrict__ q8, int dim8,
                                                               const uint32_t* __restrict__ stats, float* __restrict__ qinfo) {
  typedef T vec8 __attribute__((ext_vector_type(8)));
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q >= nq) return;
  const T* row = xq + (size_t)q * dim;
  float m = 0.f;
  for (int c = lane; c < dim / 8; c += 64) {
    const vec8 v = *(const vec8*)(row + c * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf((float)v[i]));
  }
  m = wmax(m);
  const float sq = scale_of(m), inv = 1.0f / sq;
  float e2 = 0.f, n2 = 0.f, a2 = 0.f;
  for (int c = lane; c < dim8 / 8; c += 64) {
    float x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < dim / 8) {
      const vec8 v = *(const vec8*)(row + c * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = (float)v[i];
    }
    const uint2 w = quantise8(x, sq, inv, e2, n2);
    *(uint2*)(q8 + (size_t)q * dim8 + c * 8) = w;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float r = sq * (float)(int8_t)(((i < 4 ? w.x : w.y) >> (8 * (i & 3))) & 0xFFu);
      a2 = fmaf(r, r, a2);
    }
  }
  e2 = wsum(e2);
  a2 = wsum(a2);
  n2 = wsum(n2);
  if (lane == 0) {
    const float s = scale_of(__uint_as_float(stats[0]));
    const float dmax = sqrtf(__uint_as_float(stats[1])), xmax = sqrtf(__uint_as_float(stats[2]));
    const float a = sqrtf(a2), e = sqrtf(e2);
    // 1.001: f32 rounding of the sums and square roots above; 1e-5 |q||x|: f32 rounding of the re-scored dot product
    // (<= 24 sequential terms per lane + a 6-level tree), which is what the certificate compares against
    // 2 sq s: the i32 -> f32 conversion of P is exact below 2^24 and off by at most 2 units up to 127*127*1536
    const float eps = 1.001f * (a * dmax + e * xmax) + 1e-5f * sqrtf(n2) * xmax + 2.f * sq * s;
    qinfo[2 * q] = sq * s;
    qinfo[2 * q + 1] = eps;
  }
}

// one workgroup per query: exact f32 scores of the listed rows from the f16/bf16 corpus, sort, emit top-k, certify
template <typename T>
__global__ __launch_bounds__(256) void rescore_kernel(const T* __restrict__ xb, const T* __restrict__ xq, int dim, int L,
                                                      const float* __restrict__ PL, const int64_t* __restrict__ IL,
                                                      const float* __restrict__ qinfo, int k, float* __restrict__ D,
                                                      int64_t* __restrict__ I, int64_t id_offset, uint8_t* __restrict__ exact,
                                                      const uint8_t* __restrict__ mask, int64_t mask_stride) {
  typedef T vec8 __attribute__((ext_vector_type(8)));
  __shared__ float qf[kMaxResidentDim * 2];
  __shared__ uint64_t keys[kMaxK];
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (mask && !mask[(size_t)q * mask_stride]) {
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
      D[(size_t)q * k + i] = -__builtin_inff();
      I[(size_t)q * k + i] = -1;
    }
    if (threadIdx.x == 0) exact[q] = 1;
    return;
  }
  for (int i = threadIdx.x; i < dim; i += blockDim.x) qf[i] = (float)xq[(size_t)q * dim + i];
  const int np = pow2_ceil(L < 2 ? 2 : L);
  for (int i = L + threadIdx.x; i < np; i += blockDim.x) keys[i] = 0;
  __syncthreads();
  for (int c = wave; c < L; c += 4) {
    const int64_t id = IL[(size_t)q * L + c];  // wave-uniform
    float acc = 0.f;
    if (id >= 0) {
      const T* row = xb + (size_t)id * dim;
      for (int ch = lane; ch < dim / 8; ch += 64) {
        const vec8 v = *(const vec8*)(row + ch * 8);
        const f32x4 q0 = *(const f32x4*)(qf + ch * 8), q1 = *(const f32x4*)(qf + ch * 8 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = fmaf((float)v[i], q0[i], acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = fmaf((float)v[4 + i], q1[i], acc);
      }
      acc = wsum(acc);
    }
    if (lane == 0) keys[c] = (id >= 0 && acc == acc) ? make_key(acc, (uint32_t)id) : 0ull;  // NaN rows are never returned
  }
  __syncthreads();
  bitonic_sort_desc(keys, np);
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    const uint64_t key = i < np ? keys[i] : 0ull;
    D[(size_t)q * k + i] = key ? key_score(key) : -__builtin_inff();
    I[(size_t)q * k + i] = key ? (int64_t)key_id(key) + id_offset : -1;
  }
  if (threadIdx.x == 0) {
    bool ok;
    if (IL[(size_t)q * L + L - 1] < 0) {
      ok = true;  // the list holds every row the int8 scan ranked (NaN rows excepted, which no search returns)
    } else {
      const uint64_t kth = k <= np ? keys[k - 1] : 0ull;
      const float bound = fmaf(qinfo[2 * q], PL[(size_t)q * L + L - 1], qinfo[2 * q + 1]);
      ok = kth != 0 && bound < key_score(kth);
    }
    exact[q] = ok ? 1 : 0;
  }
}

int rows_grid(int64_t n) {
  int64_t g = (n + 3) / 4;
  if (g > 8192) g = 8192;
  return g < 1 ? 1 : (int)g;
}

}  // namespace

hipError_t launch_screen_build(const void* xb, int dtype, int64_t n, int dim, int8_t* x8, int dim8, uint32_t* stats, hipStream_t st) {
  hipError_t e = hipMemsetAsync(stats, 0, 8 * sizeof(uint32_t), st);
  if (e != hipSuccess || n == 0) return e;
  const int64_t chunks = n * dim / 8;
  int64_t g = (chunks + 255) / 256;
  if (g > 4096) g = 4096;
  if (dtype == RR_DTYPE_F16) {
    hipLaunchKernelGGL(absmax_kernel<_Float16>, dim3((int)g), dim3(256), 0, st, (const _Float16*)xb, chunks, stats);
    hipLaunchKernelGGL(quantise_rows_kernel<_Float16>, dim3(rows_grid(n)), dim3(256), 0, st, (const _Float16*)xb, n, dim, x8, dim8, stats);
  } else if (dtype == RR_DTYPE_BF16) {
    hipLaunchKernelGGL(absmax_kernel<__bf16>, dim3((int)g), dim3(256), 0, st, (const __bf16*)xb, chunks, stats);
    hipLaunchKernelGGL(quantise_rows_kernel<__bf16>, dim3(rows_grid(n)), dim3(256), 0, st, (const __bf16*)xb, n, dim, x8, dim8, stats);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_screen_queries(const void* xq, int dtype, int nq, int dim, int8_t* q8, int dim8, const uint32_t* stats, float* qinfo,
                                 hipStream_t st) {
  if (nq == 0) return hipSuccess;
  const int g = (nq + 3) / 4;
  if (dtype == RR_DTYPE_F16)
    hipLaunchKernelGGL(quantise_queries_kernel<_Float16>, dim3(g), dim3(256), 0, st, (const _Float16*)xq, nq, dim, q8, dim8, stats, qinfo);
  else if (dtype == RR_DTYPE_BF16)
    hipLaunchKernelGGL(quantise_queries_kernel<__bf16>, dim3(g), dim3(256), 0, st, (const __bf16*)xq, nq, dim, q8, dim8, stats, qinfo);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_rescore(const void* xb, const void* xq, int dtype, int dim, int nq, int L, const float* PL, const int64_t* IL,
                          const float* qinfo, int k, float* D, int64_t* I, int64_t id_offset, uint8_t* exact, const uint8_t* mask,
                          int64_t mask_stride, hipStream_t st) {
  if (nq == 0) return hipSuccess;
  if (dtype == RR_DTYPE_F16)
    hipLaunchKernelGGL(rescore_kernel<_Float16>, dim3(nq), dim3(256), 0, st, (const _Float16*)xb, (const _Float16*)xq, dim, L, PL, IL,
                       qinfo, k, D, I, id_offset, exact, mask, mask_stride);
  else if (dtype == RR_DTYPE_BF16)
    hipLaunchKernelGGL(rescore_kernel<__bf16>, dim3(nq), dim3(256), 0, st, (const __bf16*)xb, (const __bf16*)xq, dim, L, PL, IL, qinfo, k,
                       D, I, id_offset, exact, mask, mask_stride);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace rr
