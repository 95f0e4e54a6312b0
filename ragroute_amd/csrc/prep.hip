// K0: row-wise L2 normalisation and the f32 -> f16/bf16 ingest conversion.
//  * l2_normalize_f32 : in-place, zero-norm rows unchanged — `faiss.normalize_L2`, reference
//                       ragroute/data_source.py:198-199
//  * rows_to_half     : f32 rows -> HBM scan format (padded leading dimension), optional normalise
// One wave per row; 16-byte loads; wave-shuffle reduction of the squared norm.
#include "rr_common.h"
#include "rr_kernels.h"

namespace rr {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float row_sumsq(const float* x, int64_t d, int lane) {
  float acc = 0.f;
  if ((d & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
    const float4* x4 = (const float4*)x;
    for (int64_t i = lane; i < d / 4; i += 64) {
      const float4 v = x4[i];
      acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  } else {
    for (int64_t i = lane; i < d; i += 64) acc += x[i] * x[i];
  }
  return wave_sum(acc);
}

__global__ __launch_bounds__(256) void l2_normalize_f32_kernel(float* x, int64_t n, int64_t d) {
  const int lane = threadIdx.x & 63;
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * wpb) {
    float* xr = x + row * d;
    const float nr = row_sumsq(xr, d, lane);
    if (nr > 0.f) {
      const float inv = 1.0f / sqrtf(nr);
      for (int64_t i = lane; i < d; i += 64) xr[i] *= inv;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rows_to_half_kernel(const float* x, int64_t n, int64_t d, int64_t ld_in, T* out,
                                                           int64_t d_out, int normalize) {
  const int lane = threadIdx.x & 63;
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * wpb) {
    const float* xr = x + row * ld_in;
    float inv = 1.f;
    if (normalize) {
      const float nr = row_sumsq(xr, d, lane);
      if (nr > 0.f) inv = 1.0f / sqrtf(nr);
    }
    T* o = out + row * d_out;
    if ((d & 3) == 0 && (d_out & 3) == 0 && ((((uintptr_t)xr) | ((uintptr_t)o << 1)) & 15) == 0) {
      // 16-byte loads, 8-byte stores (Guideline 13: hipcc does not vectorise these by itself)
      typedef T vec4 __attribute__((ext_vector_type(4)));
      const float4* x4 = (const float4*)xr;
      vec4* o4 = (vec4*)o;
      for (int64_t i = lane; i < d_out / 4; i += 64) {
        vec4 h = {(T)0.f, (T)0.f, (T)0.f, (T)0.f};
        if (i < d / 4) {
          const float4 v = x4[i];
          h = vec4{(T)(v.x * inv), (T)(v.y * inv), (T)(v.z * inv), (T)(v.w * inv)};
        }
        o4[i] = h;
      }
    } else {
      for (int64_t i = lane; i < d_out; i += 64) o[i] = i < d ? (T)(xr[i] * inv) : (T)0.f;
    }
  }
}

// |x|^2 / 2 of every stored (already rounded) row, f32: the L2 metric's per-row term.  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void half_sqnorm_kernel(const T* __restrict__ xb, int64_t n, int dim, float* __restrict__ out) {
  typedef T vec8 __attribute__((ext_vector_type(8)));
  const int lane = threadIdx.x & 63;
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * wpb) {
    float acc = 0.f;
    for (int c = lane; c < dim / 8; c += 64) {
      const vec8 v = *(const vec8*)(xb + row * dim + c * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += (float)v[i] * (float)v[i];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] = 0.5f * acc;
  }
}

// Column means of an HBM-resident corpus (the "centroid" the router consumes, reference router.py:147-151; the
// reference reads it from *_stats.json produced off-tree).  Each thread owns 8 consecutive columns (16-byte loads),
// a workgroup strides over rows, partial sums meet with float atomics (<= 2048 adds per column).
template <typename T>
__global__ __launch_bounds__(256) void column_sum_kernel(const T* __restrict__ xb, int64_t n, int dim, float* __restrict__ sums) {
  typedef T vec8 __attribute__((ext_vector_type(8)));
  const int cols8 = dim / 8;                       // dim is a multiple of 64
  const int tpr = cols8 < 256 ? cols8 : 256;       // threads that cover one row pass
  const int rows_per_pass = 256 / tpr > 0 ? 256 / tpr : 1;
  const int t = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
  for (int c8 = t; c8 < cols8; c8 += tpr) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rsub < rows_per_pass) {
      for (int64_t r = (int64_t)blockIdx.x * rows_per_pass + rsub; r < n; r += (int64_t)gridDim.x * rows_per_pass) {
        const vec8 v = *(const vec8*)(xb + r * dim + c8 * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(&sums[c8 * 8 + i], acc[i]);
    }
  }
}

__global__ void scale_kernel(float* x, int d, float s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d) x[i] *= s;
}

hipError_t launch_centroid(const void* xb, int dtype, int64_t n, int dim, int d, float* out, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out, 0, (size_t)dim * sizeof(float), st);
  if (e != hipSuccess) return e;
  if (n == 0) return hipSuccess;
  int64_t g = (n + 63) / 64;
  if (g > 2048) g = 2048;
  if (dtype == RR_DTYPE_F16) hipLaunchKernelGGL(column_sum_kernel<_Float16>, dim3((int)g), dim3(256), 0, st, (const _Float16*)xb, n, dim, out);
  else if (dtype == RR_DTYPE_BF16) hipLaunchKernelGGL(column_sum_kernel<__bf16>, dim3((int)g), dim3(256), 0, st, (const __bf16*)xb, n, dim, out);
  else return hipErrorInvalidValue;
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(scale_kernel, dim3((d + 255) / 256), dim3(256), 0, st, out, d, 1.0f / (float)n);
  return hipGetLastError();
}

static int grid_for_rows(int64_t n) {
  int64_t g = (n + 3) / 4;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

hipError_t launch_half_sqnorms(const void* xb, int dtype, int64_t n, int dim, float* out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  if (dtype == RR_DTYPE_F16) hipLaunchKernelGGL(half_sqnorm_kernel<_Float16>, dim3(grid_for_rows(n)), dim3(256), 0, st, (const _Float16*)xb, n, dim, out);
  else if (dtype == RR_DTYPE_BF16) hipLaunchKernelGGL(half_sqnorm_kernel<__bf16>, dim3(grid_for_rows(n)), dim3(256), 0, st, (const __bf16*)xb, n, dim, out);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_l2_normalize_f32(float* x, int64_t n, int64_t d, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(l2_normalize_f32_kernel, dim3(grid_for_rows(n)), dim3(256), 0, st, x, n, d);
  return hipGetLastError();
}

hipError_t launch_rows_to_half(const float* x, int64_t n, int64_t d, int64_t ld_in, void* out, int dtype, int64_t d_out,
                               int normalize, hipStream_t st) {
  if (n == 0) return hipSuccess;
  if (dtype == RR_DTYPE_F16)
    hipLaunchKernelGGL(rows_to_half_kernel<_Float16>, dim3(grid_for_rows(n)), dim3(256), 0, st, x, n, d, ld_in,
                       (_Float16*)out, d_out, normalize);
  else if (dtype == RR_DTYPE_BF16)
    hipLaunchKernelGGL(rows_to_half_kernel<__bf16>, dim3(grid_for_rows(n)), dim3(256), 0, st, x, n, d, ld_in,
                       (__bf16*)out, d_out, normalize);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace rr
