// Microbenchmark (development): how many bytes per cycle does ONE CU's vector load path move, by instruction form?
//   hipcc --offload-arch=gfx950 -O3 tools/ta_rate.hip -o /tmp/ta_rate && /tmp/ta_rate
// 256 workgroups x 256 threads (one wave per SIMD, as the scan kernels run), every wave streams 1-KiB pieces (64 lanes x 16 B,
// contiguous) out of a cache-resident buffer with a window of W pieces in flight.  Forms: global_load_dwordx4 (SGPR base +
// 32-bit lane offset), buffer_load_dwordx4 offen, the same two writing LDS directly (LDS-DMA), and global_load_dwordx2 x 2.
// Prints cycles per piece and CU (s_memtime) and bytes per cycle and CU for 1, 2 and 4 active waves.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                           \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)

constexpr int kPieces = 4096;       // per wave
constexpr int kWindow = 8;          // loads in flight per wave
constexpr uint32_t kSpan = 1 << 20; // bytes each workgroup cycles through (stays in L2)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 make_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  u32x4 r;
  r.x = (uint32_t)a;
  r.y = (uint32_t)(a >> 32) & 0xFFFF;
  r.z = bytes;
  r.w = 0x00020000;   // raw buffer, dword format (gfx9 family)
  return r;
}

template <int FORM>
__global__ __launch_bounds__(256, 1) void rate_kernel(const char* __restrict__ src, uint64_t* __restrict__ cycles, uint32_t* sink,
                                                      int active_waves) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src;   // every workgroup streams the same MiB: L2 hits after the first touch
  const u32x4 rsrc = make_rsrc(base, kSpan);
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  if (wave < active_waves) {
    uint32_t off = (uint32_t)wave * (kSpan / 4) + (uint32_t)lane * 16;
    const uint32_t lim = (uint32_t)(wave + 1) * (kSpan / 4);
    for (int i = 0; i < kPieces; ++i) {
      // Register destinations are hard-wired high VGPRs the compiler never allocates (it needs ~20): an "=v" output would be
      // free for reuse from the asm statement on, while the load is still in flight - and a later piece's OFFSET register
      // then gets overwritten by landing data (the first version of this file faulted exactly that way).
      if (FORM == 0) {
        asm volatile("global_load_dwordx4 v[64:67], %0, %1" ::"v"(off), "s"(base) : "memory", "v64", "v65", "v66", "v67");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWindow - 1) : "memory");
      } else if (FORM == 1) {
        asm volatile("buffer_load_dwordx4 v[64:67], %0, %1, 0 offen" ::"v"(off), "s"(rsrc) : "memory", "v64", "v65", "v66", "v67");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWindow - 1) : "memory");
      } else if (FORM == 2) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                         (__attribute__((address_space(3))) void*)(smem + wave * 8192 + (i & 7) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWindow - 1) : "memory");
      } else if (FORM == 3) {
        const uint32_t m0v = (uint32_t)(uintptr_t)(smem + wave * 8192 + (i & 7) * 1024) & 0xFFFF;
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(off), "s"(rsrc) : "memory", "m0");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWindow - 1) : "memory");
      } else {
        asm volatile("global_load_dwordx2 v[64:65], %0, %1\n\tglobal_load_dwordx2 v[66:67], %0, %1 offset:8" ::"v"(off), "s"(base)
                     : "memory", "v64", "v65", "v66", "v67");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kWindow - 2) : "memory");
      }
      off += 1024;
      if (off >= lim) off -= kSpan / 4;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (t1 == 0x12345678u) sink[0] = smem[lane];
}

template <int FORM>
static void run(const char* name, const char* d_src, uint64_t* d_cyc, uint32_t* d_sink) {
  for (int waves : {1, 2, 4}) {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(rate_kernel<FORM>, dim3(256), dim3(256), 32768, 0, d_src, d_cyc, d_sink, waves);
      CHECK(hipDeviceSynchronize());
      std::vector<uint64_t> c(256);
      CHECK(hipMemcpy(c.data(), d_cyc, 256 * 8, hipMemcpyDeviceToHost));
      double mean = 0;
      for (auto v : c) mean += (double)v;
      mean /= 256;
      if (mean < best) best = mean;
    }
    // s_memtime counts shader cycles (MI355X_MICROARCH.md)
    const double pieces = (double)kPieces * waves;
    printf("%-34s waves=%d  %6.1f cycles per 1-KiB piece and CU = %5.1f B per cycle and CU\n", name, waves, best / pieces,
           pieces * 1024 / best);
  }
}

int main() {
  char* d_src;
  uint64_t* d_cyc;
  uint32_t* d_sink;
  CHECK(hipMalloc(&d_src, (size_t)kSpan));
  CHECK(hipMemset(d_src, 1, (size_t)kSpan));
  CHECK(hipMalloc(&d_cyc, 256 * 8));
  CHECK(hipMalloc(&d_sink, 64));
  CHECK(hipFuncSetAttribute((const void*)rate_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
  CHECK(hipFuncSetAttribute((const void*)rate_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
  run<0>("global_load_dwordx4 (saddr)", d_src, d_cyc, d_sink);
  run<1>("buffer_load_dwordx4 offen", d_src, d_cyc, d_sink);
  run<2>("global_load_lds_dwordx4", d_src, d_cyc, d_sink);
  run<3>("buffer_load_dwordx4 offen lds", d_src, d_cyc, d_sink);
  run<4>("global_load_dwordx2 x 2", d_src, d_cyc, d_sink);
  return 0;
}
