/* oracle/flat_oracle.c — TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C CPU restatement of the numeric call sites on RAGRoute's retrieval hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (ragroute_amd/) never does.
 *
 * PARITY UNPINNED for the search arithmetic: the reference delegates `index.search` to the
 * third-party wheel faiss-cpu==1.7.4 (environment.yml:55), which is neither vendored under
 * /root/reference nor installed here, and the reference holds no golden vectors for it.  What is
 * restated is FAISS's published IndexFlatIP contract as the reference uses it:
 *   - call shape  D[nq,k] f32, I[nq,k] i64 = index.search(xq f32[nq,d], k)   data_source.py:158,186,203
 *   - single f32 query row per call                                          data_source.py:113-114
 *   - higher score = better (merge sorts descending)                         rerank.py:5
 *   - exhaustive and exact; results best-first; k > ntotal pads (-inf, -1)
 *   - faiss.normalize_L2: x /= ||x||_2 per row in f32, zero rows untouched   data_source.py:198-199
 * Ties (unspecified by FAISS's heap) are defined here as: ascending row id.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* a is better than b: higher score, then lower id */
static int better(float sa, int64_t ia, float sb, int64_t ib) { return sa > sb || (sa == sb && ia < ib); }

/* insert (s,i) into a best-first sorted list of length *n <= k */
static void topk_insert(float* D, int64_t* I, int* n, int k, float s, int64_t i) {
  if (s != s) return; /* NaN never selected */
  if (*n == k && !better(s, i, D[k - 1], I[k - 1])) return;
  int p = *n < k ? (*n)++ : k - 1;
  while (p > 0 && better(s, i, D[p - 1], I[p - 1])) { D[p] = D[p - 1]; I[p] = I[p - 1]; --p; }
  D[p] = s; I[p] = i;
}

/* index.search for a flat inner-product index (data_source.py:158,186,203).
 * Scores: f32 rounding of a double-precision dot product of the f32 inputs. */
void oracle_flat_search_ip(const float* xb, int64_t n, int d, const float* xq, int nq, int k, float* D, int64_t* I) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int q = 0; q < nq; ++q) {
    float* Dq = D + (size_t)q * k; int64_t* Iq = I + (size_t)q * k;
    int cnt = 0;
    const float* x = xq + (size_t)q * d;
    for (int64_t r = 0; r < n; ++r) {
      const float* y = xb + (size_t)r * d;
      double acc = 0.0;
      for (int j = 0; j < d; ++j) acc += (double)x[j] * (double)y[j];
      topk_insert(Dq, Iq, &cnt, k, (float)acc, r);
    }
    for (int j = cnt; j < k; ++j) { Dq[j] = -INFINITY; Iq[j] = -1; }
  }
}

/* Flat squared-L2 index (faiss.IndexFlatL2 contract): k nearest rows, nearest first, D = |q - x|^2 (f32 rounding of an f64
 * sum), ties by ascending id, padding (+inf, -1). */
void oracle_flat_search_l2(const float* xb, int64_t n, int d, const float* xq, int nq, int k, float* D, int64_t* I) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int q = 0; q < nq; ++q) {
    float* Dq = D + (size_t)q * k; int64_t* Iq = I + (size_t)q * k;
    int cnt = 0;
    const float* x = xq + (size_t)q * d;
    for (int64_t r = 0; r < n; ++r) {
      const float* y = xb + (size_t)r * d;
      double acc = 0.0;
      for (int j = 0; j < d; ++j) { const double t = (double)x[j] - (double)y[j]; acc += t * t; }
      topk_insert(Dq, Iq, &cnt, k, -(float)acc, r);
    }
    for (int j = 0; j < cnt; ++j) Dq[j] = -Dq[j];
    for (int j = cnt; j < k; ++j) { Dq[j] = INFINITY; Iq[j] = -1; }
  }
}

/* The reference's actual call pattern: ONE query per call (data_source.py:114), f32 arithmetic
 * as a CPU flat index does it, rows split over the host cores.  Used as bench.py's cpu_baseline. */
void oracle_flat_search_ip_f32_single(const float* xb, int64_t n, int d, const float* xq, int k, float* D, int64_t* I) {
  int nt = oracle_num_threads();
  float* Dt = (float*)malloc((size_t)nt * k * sizeof(float));
  int64_t* It = (int64_t*)malloc((size_t)nt * k * sizeof(int64_t));
  int* ct = (int*)calloc(nt, sizeof(int));
#pragma omp parallel
  {
#ifdef _OPENMP
    int t = omp_get_thread_num();
#else
    int t = 0;
#endif
    float* Dq = Dt + (size_t)t * k; int64_t* Iq = It + (size_t)t * k; int cnt = 0;
#pragma omp for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
      const float* y = xb + (size_t)r * d;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int j = 0;
      for (; j + 3 < d; j += 4) { a0 += xq[j] * y[j]; a1 += xq[j + 1] * y[j + 1]; a2 += xq[j + 2] * y[j + 2]; a3 += xq[j + 3] * y[j + 3]; }
      for (; j < d; ++j) a0 += xq[j] * y[j];
      topk_insert(Dq, Iq, &cnt, k, (a0 + a1) + (a2 + a3), r);
    }
    ct[t] = cnt;
  }
  int cnt = 0;
  for (int t = 0; t < nt; ++t)
    for (int j = 0; j < ct[t]; ++j) topk_insert(D, I, &cnt, k, Dt[(size_t)t * k + j], It[(size_t)t * k + j]);
  for (int j = cnt; j < k; ++j) { D[j] = -INFINITY; I[j] = -1; }
  free(Dt); free(It); free(ct);
}

/* faiss.normalize_L2 (data_source.py:198-199): f32, rows with zero norm left unchanged */
void oracle_normalize_l2(float* x, int64_t n, int64_t d) {
  for (int64_t r = 0; r < n; ++r) {
    float* v = x + r * d;
    float nr = 0.f;
    for (int64_t j = 0; j < d; ++j) nr += v[j] * v[j];
    if (nr > 0.f) {
      const float inv = 1.0f / sqrtf(nr);
      for (int64_t j = 0; j < d; ++j) v[j] *= inv;
    }
  }
}

/* Cross-source merge (rerank.py:3-9 descending, :28-34 ascending) of m (score,id) candidates per
 * query; ties by ascending id; id < 0 or NaN = padding. */
void oracle_merge_topk(const float* Din, const int64_t* Iin, int nq, int m, int k, int descending, float* Dout, int64_t* Iout) {
  for (int q = 0; q < nq; ++q) {
    float* Dq = Dout + (size_t)q * k; int64_t* Iq = Iout + (size_t)q * k;
    int cnt = 0;
    for (int c = 0; c < m; ++c) {
      float s = Din[(size_t)q * m + c]; int64_t i = Iin[(size_t)q * m + c];
      if (i < 0) continue;
      topk_insert(Dq, Iq, &cnt, k, descending ? s : -s, i);
    }
    if (!descending) for (int j = 0; j < cnt; ++j) Dq[j] = -Dq[j];
    for (int j = cnt; j < k; ++j) { Dq[j] = descending ? -INFINITY : INFINITY; Iq[j] = -1; }
  }
}
