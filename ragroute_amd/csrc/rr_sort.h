// Workgroup-wide bitonic sort of 64-bit order keys in LDS (shared by select.hip and screen.hip).
#pragma once
#include "rr_common.h"

namespace rr {

__device__ __forceinline__ int pow2_ceil(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

// descending bitonic sort of n (power of two) u64 keys in LDS by the whole workgroup
__device__ inline void bitonic_sort_desc(uint64_t* s, int n) {
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

}  // namespace rr
