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

// ---- batched form on the matrix cores (round 2) ----------------------------------------------------------------------------
// The kernel above re-reads W1q' once per 4 queries (FeB4RAG, 256 queries: 64 x 10.5 MB through L2) and runs fc2 as one wave
// per (query, source) row: 157 us per 256 FeB4RAG queries, latency-bound on L2.  For batches of >= 32 queries the two GEMM-shaped
// layers go to the f32 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate - exactly an fmaf chain - at the f32 vector
// rate, but 64 queries share every weight load):
//   router_fc1_kernel   grid (ceil(nq/64), n_models, d_max/128): 64 queries x 256 outputs x one 128-wide K chunk per workgroup;
//                       A = the queries' chunk in LDS (row stride 129: conflict-free), B = W1q' rows streamed from L2 with one
//                       16-byte load per lane and K pair (columns interleaved 4 n + i over the wave's four MFMAs), partial sums
//                       to the workspace.  A chunk whose inputs are all zero (zero padding of narrow encoders, router.py:245-249)
//                       is skipped and flagged.
//   router_head_kernel  grid (ceil(nq/32), n_sources): 32 (query, source) rows per workgroup: fc1 = valid partials in chunk order
//                       + c1[c], LayerNorm, ReLU -> LDS; fc2 on the matrix cores (each wave 32 rows x 32 outputs); LayerNorm,
//                       ReLU, fc3, sigmoid, threshold.  Sums are taken in a fixed order: results do not depend on timing.
typedef float f32x16v __attribute__((ext_vector_type(16)));
constexpr int kFc1Chunk = 128;     // K per fc1 workgroup
constexpr int kFc1Q = 64;          // queries per fc1 workgroup
constexpr int kHeadRows = 32;      // (query, source) rows per head workgroup

__global__ __launch_bounds__(256) void router_fc1_kernel(rr_router_weights w, const float* __restrict__ xq, int nq,
                                                         float* __restrict__ part, uint8_t* __restrict__ valid) {
  __shared__ float xs[kFc1Q][kFc1Chunk + 1];
  __shared__ int any_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x, m = blockIdx.y, ch = blockIdx.z;
  const int n_chunks = gridDim.z;
  const int dmax = w.d_max;
  const int k0 = ch * kFc1Chunk;
  if (tid == 0) any_s = 0;
  __syncthreads();
  bool any = false;
  if ((dmax & 3) == 0) {
    // 8 unconditional 16-byte loads per thread, all in flight together (clamped addresses, invalid positions zeroed afterwards)
    float4 t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i, r = idx >> 5, k = (idx & 31) * 4;
      const int q = min(qt * kFc1Q + r, nq - 1), kc = min(k0 + k, dmax - 4);
      t[i] = *(const float4*)(xq + ((size_t)q * w.n_models + m) * dmax + kc);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i, r = idx >> 5, k = (idx & 31) * 4;
      const bool ok = qt * kFc1Q + r < nq && k0 + k < dmax;
      const float4 v = ok ? t[i] : float4{0.f, 0.f, 0.f, 0.f};
      xs[r][k] = v.x; xs[r][k + 1] = v.y; xs[r][k + 2] = v.z; xs[r][k + 3] = v.w;
      any = any || (v.x != 0.f) || (v.y != 0.f) || (v.z != 0.f) || (v.w != 0.f);   // NaN != 0: a NaN input still reaches fc1
    }
  } else {
    for (int i = tid; i < kFc1Q * kFc1Chunk; i += 256) {
      const int r = i / kFc1Chunk, k = i - r * kFc1Chunk;
      const int q = qt * kFc1Q + r;
      const float v = (q < nq && k0 + k < dmax) ? xq[((size_t)q * w.n_models + m) * dmax + k0 + k] : 0.f;
      xs[r][k] = v;
      any = any || (v != 0.f);
    }
  }
  if (any) any_s = 1;
  __syncthreads();
  const bool live = any_s != 0;
  if (tid == 0) valid[((size_t)qt * w.n_models + m) * n_chunks + ch] = live ? 1 : 0;
  if (!live) return;
  // wave (wq, wc): queries 32 wq .. +31, outputs 128 wc + 4 n + i  (n = lane & 31, i = MFMA 0..3)
  const int wq = wave >> 1, wc = wave & 1;
  const int n = lane & 31, kk = lane >> 5;
  f32x16v acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  // Operands are fetched DEPTH K pairs ahead with UNCONDITIONAL loads (a predicated load makes hipcc fall back to
  // s_waitcnt vmcnt(0) and drain the prefetch): past the end of the chunk, or of d_max, the row index is clamped to a valid
  // row - the matching xs entries are zero, so the product contributes nothing (weights are finite).
  const float* wcol = w.w1q + 128 * wc + 4 * n;
  const int krow_max = dmax - 1 - k0;              // last valid row of this chunk, relative to k0
  constexpr int NKP = kFc1Chunk / 2, DEPTH = 8;
  float4 b[DEPTH];
  float av[DEPTH];
  auto fetch = [&](int kp, float4& bo, float& ao) {
    const int kpc = kp < NKP ? kp : NKP - 1;
    const int kr = min(2 * kpc + kk, krow_max);
    bo = *(const float4*)(wcol + (size_t)(k0 + kr) * 256);
    ao = xs[32 * wq + n][2 * kpc + kk];
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch(d, b[d], av[d]);
#pragma unroll 1
  for (int kp0 = 0; kp0 < NKP; kp0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const float a = av[d];
      const float4 bv = b[d];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv.w, acc[3], 0, 0, 0);
      fetch(kp0 + d + DEPTH, b[d], av[d]);
      __builtin_amdgcn_sched_barrier(0);   // keep the loads where they are written: hipcc otherwise sinks them to 2 in flight
    }
  }
  // D layout of the 32x32 MFMA: element e of lane l is row (e & 3) + 8 (e >> 2) + 4 (l >> 5), column l & 31
  float* pp = part + ((((size_t)qt * w.n_models + m) * n_chunks + ch) * kFc1Q) * 256;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int r = 32 * wq + (e & 3) + 8 * (e >> 2) + 4 * kk;
    *(float4*)(pp + (size_t)r * 256 + 128 * wc + 4 * n) = float4{acc[0][e], acc[1][e], acc[2][e], acc[3][e]};
  }
}

__global__ __launch_bounds__(256) void router_head_kernel(rr_router_weights w, int nq, int n_chunks, const float* __restrict__ part,
                                                          const uint8_t* __restrict__ valid, float* __restrict__ logits,
                                                          uint8_t* __restrict__ mask) {
  __shared__ float h1[kHeadRows][256 + 1];
  __shared__ float z[kHeadRows][128 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.x * kHeadRows, c = blockIdx.y;
  const int m = w.model_of_source[c];
  // the 32 rows of this workgroup belong to one 64-query tile: its list of live K chunks (in chunk order: a fixed summation order)
  __shared__ int vlist[64], nvalid_s;
  const int qt = q0 / kFc1Q;
  if (wave == 0) {
    const uint8_t* vp = valid + ((size_t)qt * w.n_models + m) * n_chunks;
    const bool live = lane < n_chunks && vp[lane] != 0;     // d_max <= 8192 -> at most 64 chunks
    const uint64_t bal = __builtin_amdgcn_ballot_w64(live);
    if (live) vlist[__builtin_popcountll(bal & ((1ull << lane) - 1))] = lane;
    if (lane == 0) nvalid_s = __builtin_popcountll(bal);
  }
  __syncthreads();
  const int nvalid = nvalid_s;
  // fc1 rows: 8 rows per wave, 4 outputs per lane; chunk loop outermost so that the 32 loads of a chunk are in flight together
  {
    float v[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[r][i] = 0.f;
    const int qr0 = q0 - qt * kFc1Q + wave * 8;
    const float* pbase = part + (((size_t)qt * w.n_models + m) * n_chunks * kFc1Q + qr0) * 256 + lane;
#pragma unroll 2
    for (int ci = 0; ci < nvalid; ++ci) {
      const float* p = pbase + (size_t)vlist[ci] * kFc1Q * 256;
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[r][i] += p[r * 256 + 64 * i];   // rows past nq hold zeros (their queries were loaded as 0)
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[r][i] += w.c1[c * 256 + lane + 64 * i]; s += v[r][i]; }
      const float mean = wsum(s) * (1.f / 256.f);
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[r][i] -= mean; sq += v[r][i] * v[r][i]; }
      const float rstd = 1.0f / sqrtf(wsum(sq) * (1.f / 256.f) + w.ln_eps);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
        h1[wave * 8 + r][j] = fmaxf(v[r][i] * rstd * w.ln1_g[j] + w.ln1_b[j], 0.f);
      }
    }
  }
  __syncthreads();
  // fc2: wave -> outputs 32 wave .. +31 for the 32 rows; A = h1 (row n, k = 2 kp + kk), B = W2[k][32 wave + n]
  {
    const int n = lane & 31, kk = lane >> 5;
    f32x16v acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* w2 = w.w2 + (size_t)kk * 128 + 32 * wave + n;
    constexpr int DEPTH = 16;
    float b[DEPTH], av[DEPTH];
    auto fetch = [&](int kp, float& bo, float& ao) {   // unconditional (clamped) loads, see router_fc1_kernel
      const int kpc = kp < 128 ? kp : 127;
      bo = w2[(size_t)(2 * kpc) * 128];
      ao = h1[n][2 * kpc + kk];
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(d, b[d], av[d]);
#pragma unroll 1
    for (int kp0 = 0; kp0 < 128; kp0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[d], b[d], acc, 0, 0, 0);
        fetch(kp0 + d + DEPTH, b[d], av[d]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const float bias = w.b2[32 * wave + n];
#pragma unroll
    for (int e = 0; e < 16; ++e) z[(e & 3) + 8 * (e >> 2) + 4 * kk][32 * wave + n] = acc[e] + bias;
  }
  __syncthreads();
  for (int r = wave * 8; r < wave * 8 + 8; ++r) {
    const int q = q0 + r;
    if (q >= nq) continue;                       // wave-uniform
    float a0 = z[r][lane], a1 = z[r][lane + 64];
    const float mean2 = wsum(a0 + a1) * (1.f / 128.f);
    a0 -= mean2; a1 -= mean2;
    const float rstd2 = 1.0f / sqrtf(wsum(a0 * a0 + a1 * a1) * (1.f / 128.f) + w.ln_eps);
    const float g0 = fmaxf(a0 * rstd2 * w.ln2_g[lane] + w.ln2_b[lane], 0.f);
    const float g1 = fmaxf(a1 * rstd2 * w.ln2_g[lane + 64] + w.ln2_b[lane + 64], 0.f);
    const float logit = wsum(g0 * w.w3[lane] + g1 * w.w3[lane + 64]) + w.b3;
    if (lane == 0) {
      logits[(size_t)q * w.n_sources + c] = logit;
      const float p = 1.0f / (1.0f + expf(-logit));
      mask[(size_t)q * w.n_sources + c] = p > w.prob_threshold ? 1 : 0;
    }
  }
}

static int fc1_chunks(int d_max) { return (d_max + kFc1Chunk - 1) / kFc1Chunk; }
static int fc1_tiles(int nq) { return (nq + kFc1Q - 1) / kFc1Q; }

size_t router_workspace_bytes(const rr_router_weights* w, int nq) {
  // the matrix-core form pays off for batches and for routers with several encoders or many sources (measured, 256 queries:
  // feb4rag 13 sources / 8 encoders and wikipedia 10 sources / 1 encoder faster, medrag 4 sources / 1 encoder not)
  if (nq < kRouterMfmaMinQueries || (w->n_models == 1 && w->n_sources < 8)) return 0;
  if (fc1_chunks(w->d_max) > 64) return 0;   // router_head_kernel lists the live chunks of a tile with one wave ballot
  const size_t slots = (size_t)fc1_tiles(nq) * w->n_models * fc1_chunks(w->d_max);
  return slots * kFc1Q * 256 * sizeof(float) + ((slots + 255) / 256) * 256;
}

hipError_t launch_router_mlp_ws(const rr_router_weights* w, const float* xq, int nq, float* logits, uint8_t* mask, void* ws,
                                hipStream_t st) {
  if (nq == 0) return hipSuccess;
  const int n_chunks = fc1_chunks(w->d_max), tiles = fc1_tiles(nq);
  const size_t slots = (size_t)tiles * w->n_models * n_chunks;
  float* part = (float*)ws;
  uint8_t* valid = (uint8_t*)ws + slots * kFc1Q * 256 * sizeof(float);
  hipLaunchKernelGGL(router_fc1_kernel, dim3(tiles, w->n_models, n_chunks), dim3(256), 0, st, *w, xq, nq, part, valid);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(router_head_kernel, dim3((nq + kHeadRows - 1) / kHeadRows, w->n_sources), dim3(256), 0, st, *w, nq, n_chunks, part,
                     valid, logits, mask);
  return hipGetLastError();
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
