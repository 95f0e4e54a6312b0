// K3: fused router MLP — CorpusRoutingNN forward for a batch of queries with the feature build
// (zero-pad, centroid concat, one-hot) and StandardScaler folded into fc1.
// Replaces reference ragroute/router.py:241-283 and :50-55 for nq queries at once.
//
//   fc1(x)  = W1q' q_m + c1[c]         (W1q' = fc1 query block / scale; c1 = bias + centroid/one-hot/mean terms)
//   h1      = relu(LayerNorm256(fc1))  h2 = relu(LayerNorm128(W2 h1 + b2))  logit = w3.h2 + b3
//
// Grid (ceil(nq/4), n_models), 1024 threads; W1q' is read once per 4 queries, coalesced; query values are
// broadcast from LDS; both LayerNorms and fc3 are wave-shuffle reductions.  All arithmetic f32.
#include "rr_common.h"
#include "rr_kernels.h"

namespace rr {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 1024 threads = 16 waves.  Phase 1: thread (ks, j) accumulates fc1 output j over the k-quarter ks for the 4 queries
// of the workgroup (4x the loads in flight of a 256-thread version; the loop is latency-bound on L2), partial sums
// meet in LDS.  Phase 2: one wave per (query, source) row, see below.
__global__ __launch_bounds__(1024) void router_mlp_kernel(rr_router_weights w, const float* __restrict__ xq, int nq,
                                                          float* __restrict__ logits, uint8_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int dmax = w.d_max;
  float* xs = lds;                 // [4][dmax]
  float* up = lds + 4 * dmax;      // [4 ks][4 t][256] fc1 partials
  float* hs = up + 16 * 256;       // [16 waves][256] h1 of the (query, source) row a wave is working on
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.y;
  const int q0 = blockIdx.x * 4;

  // Embeddings are zero padded to d_max (router.py:245-249): FeB4RAG pads 768- and 1024-wide models to 4096.  Zeros add
  // nothing to fc1, so the k loop stops at the last non-zero input of the workgroup's 4 queries (bit-identical sums).
  __shared__ int k_end_s;
  if (tid == 0) k_end_s = 0;
  __syncthreads();
  int last = 0;
  for (int i = tid; i < 4 * dmax; i += 1024) {
    const int t = i / dmax, k = i - t * dmax;
    const int q = q0 + t;
    const float v = q < nq ? xq[((size_t)q * w.n_models + m) * dmax + k] : 0.f;
    xs[i] = v;
    if (v != 0.f) last = k + 1;   // (NaN != 0 is true: a NaN input still reaches fc1)
  }
  if (last) atomicMax(&k_end_s, last);
  __syncthreads();
  {
    const int ks = tid >> 8, j = tid & 255;
    const int kend = k_end_s;
    const int kq = (kend + 3) / 4;
    const int k0 = ks * kq, k1 = min(kend, k0 + kq);
    float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
    const float* wp = w.w1q + j;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) {
      const float wk = wp[(size_t)k * 256];
      u0 = fmaf(xs[k], wk, u0);
      u1 = fmaf(xs[dmax + k], wk, u1);
      u2 = fmaf(xs[2 * dmax + k], wk, u2);
      u3 = fmaf(xs[3 * dmax + k], wk, u3);
    }
    float* o = up + ks * 1024 + j;
    o[0] = u0; o[256] = u1; o[512] = u2; o[768] = u3;
  }
  __syncthreads();

  // Phase 2: wave (t, slot) owns query t and every 4th source of this model; LayerNorm256, fc2 (each lane two of the 128
  // outputs over all 256 inputs), LayerNorm128, fc3 run inside the wave - no workgroup barrier, 16 (query, source) rows in flight.
  const int t = wave & 3, slot = wave >> 2;
  const int q = q0 + t;
  const bool active = q < nq;  // wave-uniform
  float* hw = hs + wave * 256;
  int ordinal = 0;
  for (int c = 0; c < w.n_sources; ++c) {
    if (w.model_of_source[c] != m) continue;  // workgroup-uniform
    if ((ordinal++ & 3) != slot || !active) continue;
    float v[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = lane + 64 * i;
      v[i] = ((up[t * 256 + j] + up[1024 + t * 256 + j]) + (up[2048 + t * 256 + j] + up[3072 + t * 256 + j])) + w.c1[c * 256 + j];
      s += v[i];
    }
    const float mean = wsum(s) * (1.f / 256.f);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] -= mean; sq += v[i] * v[i]; }
    const float rstd = 1.0f / sqrtf(wsum(sq) * (1.f / 256.f) + w.ln_eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = lane + 64 * i;
      hw[j] = fmaxf(v[i] * rstd * w.ln1_g[j] + w.ln1_b[j], 0.f);
    }
    __builtin_amdgcn_wave_barrier();  // hw is written and read by this wave only; LDS operations of one wave complete in order
    float a0 = 0.f, a1 = 0.f, c0 = 0.f, c1v = 0.f;
    const float* w2 = w.w2 + lane;
#pragma unroll 16
    for (int i = 0; i < 256; i += 2) {   // two independent chains per output; 64 loads in flight per lane (the loop is L2-latency bound)
      const float h0 = hw[i], h1 = hw[i + 1];
      a0 = fmaf(h0, w2[i * 128], a0);
      a1 = fmaf(h0, w2[i * 128 + 64], a1);
      c0 = fmaf(h1, w2[(i + 1) * 128], c0);
      c1v = fmaf(h1, w2[(i + 1) * 128 + 64], c1v);
    }
    a0 = (a0 + c0) + w.b2[lane];
    a1 = (a1 + c1v) + w.b2[lane + 64];
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
    __builtin_amdgcn_wave_barrier();  // the next source of this wave rewrites hw
  }
}

hipError_t launch_router_mlp(const rr_router_weights* w, const float* xq, int nq, float* logits, uint8_t* mask,
                             hipStream_t st) {
  if (nq == 0) return hipSuccess;
  const size_t lds = (size_t)(4 * w->d_max + 16 * 256 + 16 * 256) * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)router_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(router_mlp_kernel, dim3((nq + 3) / 4, w->n_models), dim3(1024), lds, st, *w, xq, nq, logits, mask);
  return hipGetLastError();
}

}  // namespace rr
