// Shared device/host helpers for the ragroute_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace rr {

// Tuning / diagnostic variables (RR_SCAN_VARIANT, RR_WIDE_*, RR_CHUNK_GROWTH, RR_SAMPLE_ROWS, RR_SCAN_TIMELINE, ...) re-route kernels
// and schedules.  Only development builds (-DRR_DEV_VARIANTS, ragroute_amd/_build.py) honour them: a PRODUCT build reads none, so
// a stray variable in a service's environment cannot change the kernel under test or in production
// (tests/test_cabi_exports.py::test_product_library_ignores_tuning_variables).
inline const char* tuning_env(const char* name) {
#ifdef RR_DEV_VARIANTS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kQueriesPerBlock = 256;  // one scan launch serves 256 queries (4 waves x 64)
constexpr int kTileRows = 32;          // corpus rows per MFMA tile
constexpr int kMaxK = 1024;
constexpr int kMaxResidentDim = 768;   // largest dim whose 256 queries fit the register file
constexpr int kMaxDim = 8192;
constexpr int kSelectCap = 8192;       // u64 keys sorted in LDS by the select kernels (64 KiB)
constexpr int kSampleRows = 8192;      // bootstrap sample rows (256 tiles)
constexpr int kDenseMaxRows = 8192;    // corpora up to this size take the dense path
constexpr int kChunkGrowth = 8;
constexpr int kDtypeI8 = 2;            // internal: the int8 screening copy (after RR_DTYPE_F16 / RR_DTYPE_BF16)

// ---- total order on candidates: score descending, then id ascending ---------------------
// key = ord(score) << 32 | (0xFFFFFFFF - id); larger key = better. key 0 = empty slot
// (ord(NaN) = 0, so NaN-scored rows are never selected).
__host__ __device__ inline uint32_t ord_f32(float s) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t b = __float_as_uint(s);
#else
  union { float f; uint32_t u; } cv; cv.f = s; uint32_t b = cv.u;
#endif
  if (s != s) return 0u;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float unord_f32(uint32_t o) {
  uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(b);
#else
  union { float f; uint32_t u; } cv; cv.u = b; return cv.f;
#endif
}
__host__ __device__ inline uint64_t make_key(float s, uint32_t id) {
  return ((uint64_t)ord_f32(s) << 32) | (uint64_t)(0xFFFFFFFFu - id);
}
__host__ __device__ inline float key_score(uint64_t k) { return unord_f32((uint32_t)(k >> 32)); }
__host__ __device__ inline uint32_t key_id(uint64_t k) { return 0xFFFFFFFFu - (uint32_t)k; }

// largest float strictly below x (x finite or -inf); s > next_below(t)  <=>  s >= t
__host__ __device__ inline float next_below(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (x == -__builtin_inff()) return x;
  uint32_t b = __float_as_uint(x);
  if ((b & 0x7FFFFFFFu) == 0) return __uint_as_float(0x80000001u);
  return __uint_as_float((b & 0x80000000u) ? b + 1 : b - 1);
#else
  union { float f; uint32_t u; } cv; cv.f = x;
  if (x < 0 && x * 2 == x) return x;
  if ((cv.u & 0x7FFFFFFFu) == 0) { cv.u = 0x80000001u; return cv.f; }
  cv.u = (cv.u & 0x80000000u) ? cv.u + 1 : cv.u - 1;
  return cv.f;
#endif
}

// ---- segmented search: which tiles of a segment a chunk launch covers ------------------------------------------------------
// Every chunk takes the next slice of EVERY segment - tiles [seg_cut(t, frac[c-1]), seg_cut(t, frac[c])) of a segment of t tiles,
// frac a 2^-32 fixed-point fraction growing geometrically to 1 - so that a query routed to any subset of the segments sees its
// own rows arrive in geometrically growing portions (the schedule the exact-selection cost model assumes) whatever the router
// selected.  Cuts are whole 256-row groups (the wide-row kernels' unit).  Host (launch sizes) and device (prep_kernel's range
// tables) evaluate the same integer expression.
__host__ __device__ inline uint32_t seg_cut(uint32_t tiles, uint32_t frac32) {
  if (frac32 == 0xFFFFFFFFu) return tiles;
  return (uint32_t)((((uint64_t)tiles * frac32) >> 32) & ~7ull);
}

// candidate-buffer capacity per (workgroup, query, lane-half) for a given k
__host__ __device__ inline int cand_cap_for_k(int k) {
  int c = 2 * k;
  if (c < k + 32) c = k + 32;
  return (c + 63) & ~63;
}

}  // namespace rr
