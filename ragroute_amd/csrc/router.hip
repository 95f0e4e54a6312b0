// K3: fused router MLP — CorpusRoutingNN forward for a batch of queries with the feature build
// (zero-pad, centroid concat, one-hot) and StandardScaler folded into fc1.
// Replaces reference ragroute/router.py:241-283 and :50-55 for nq queries at once.
//
//   fc1(x)  = W1q' q_m + c1[c]         (W1q' = fc1 query block / scale; c1 = bias + centroid/one-hot/mean terms)
//   h1      = relu(LayerNorm256(fc1))  h2 = relu(LayerNorm128(W2 h1 + b2))  logit = w3.h2 + b3
//
// Grid (ceil(nq/4), n_models), 256 threads.  Phase 1: thread j owns fc1 output j for 4 queries
// (W1q' read once per 4 queries, coalesced; query values broadcast from LDS).  Phase 2: wave t
// owns query t and walks the sources that use this model; both LayerNorms and fc3 are
// wave-shuffle reductions, fc2 reads h1 from a wave-private LDS row.  All arithmetic f32.
#include "rr_common.h"
#include "rr_kernels.h"

namespace rr {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void router_mlp_kernel(rr_router_weights w, const float* __restrict__ xq, int nq,
                                                         float* __restrict__ logits, uint8_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int dmax = w.d_max;
  float* xs = lds;              // [4][dmax]
  float* us = lds + 4 * dmax;   // [4][256]
  float* hs = us + 4 * 256;     // [4][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.y;
  const int q0 = blockIdx.x * 4;

  for (int i = tid; i < 4 * dmax; i += 256) {
    const int t = i / dmax, k = i - t * dmax;
    const int q = q0 + t;
    xs[i] = q < nq ? xq[((size_t)q * w.n_models + m) * dmax + k] : 0.f;
  }
  __syncthreads();
  float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
  const float* wp = w.w1q + tid;
#pragma unroll 4
  for (int k = 0; k < dmax; ++k) {
    const float wk = wp[(size_t)k * 256];
    u0 = fmaf(xs[k], wk, u0);
    u1 = fmaf(xs[dmax + k], wk, u1);
    u2 = fmaf(xs[2 * dmax + k], wk, u2);
    u3 = fmaf(xs[3 * dmax + k], wk, u3);
  }
  us[tid] = u0; us[256 + tid] = u1; us[512 + tid] = u2; us[768 + tid] = u3;
  __syncthreads();

  const int q = q0 + wave;
  if (q >= nq) return;
  const float* ut = us + wave * 256;
  float* ht = hs + wave * 256;
  for (int c = 0; c < w.n_sources; ++c) {
    if (w.model_of_source[c] != m) continue;
    float v[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = ut[lane + 64 * i] + w.c1[c * 256 + lane + 64 * i]; s += v[i]; }
    const float mean = wsum(s) * (1.f / 256.f);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] -= mean; sq += v[i] * v[i]; }
    const float rstd = 1.0f / sqrtf(wsum(sq) * (1.f / 256.f) + w.ln_eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = lane + 64 * i;
      ht[j] = fmaxf(v[i] * rstd * w.ln1_g[j] + w.ln1_b[j], 0.f);
    }
    // wave-private row: LDS writes of this wave are visible to its own later reads
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    float a0 = w.b2[lane], a1 = w.b2[lane + 64];
    for (int i = 0; i < 256; ++i) {
      const float hv = ht[i];
      a0 = fmaf(hv, w.w2[i * 128 + lane], a0);
      a1 = fmaf(hv, w.w2[i * 128 + lane + 64], a1);
    }
    const float mean2 = wsum(a0 + a1) * (1.f / 128.f);
    a0 -= mean2; a1 -= mean2;
    const float rstd2 = 1.0f / sqrtf(wsum(a0 * a0 + a1 * a1) * (1.f / 128.f) + w.ln_eps);
    const float h0 = fmaxf(a0 * rstd2 * w.ln2_g[lane] + w.ln2_b[lane], 0.f);
    const float h1 = fmaxf(a1 * rstd2 * w.ln2_g[lane + 64] + w.ln2_b[lane + 64], 0.f);
    const float logit = wsum(h0 * w.w3[lane] + h1 * w.w3[lane + 64]) + w.b3;
    if (lane == 0) {
      logits[(size_t)q * w.n_sources + c] = logit;
      const float p = 1.0f / (1.0f + expf(-logit));
      mask[(size_t)q * w.n_sources + c] = p > w.prob_threshold ? 1 : 0;
    }
  }
}

hipError_t launch_router_mlp(const rr_router_weights* w, const float* xq, int nq, float* logits, uint8_t* mask,
                             hipStream_t st) {
  if (nq == 0) return hipSuccess;
  const size_t lds = (size_t)(4 * w->d_max + 2 * 4 * 256) * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)router_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(router_mlp_kernel, dim3((nq + 3) / 4, w->n_models), dim3(256), lds, st, *w, xq, nq, logits, mask);
  return hipGetLastError();
}

}  // namespace rr
