// K1 for wide rows (d > 1536, and more than 128 queries or the L2 metric at 768 < d <= 1536).
// Replaces the arithmetic inside `index.search(query_embed, k)` (reference ragroute/data_source.py:158,186,203) for
// FeB4RAG's 1024- and 4096-wide encoders (config.py:45-57, 92-96).  Split from flat_scan.hip (the query-resident kernels).
// Kernels in this file (launch_scan_wide_t picks one):
//   flat_scan_wide_rs_kernel  209 ... 256 queries (193 ... 256 at d > 2048), either metric, any k (round 3; L2 / k > 128 / 193+ round 4):
//                             8 waves = 2 row halves x 4 query quarters, the query columns of a K step DMA'd into LDS once per CU
//   flat_scan_wide8_kernel    d <= 2048, up to 208 queries: 8 waves, queries streamed into registers
//   flat_scan_wide_pd_kernel  d > 2048 up to 192 queries, 9 / 11 / 12 query blocks at d <= 2048, every dense (sample) launch: 4 waves
//   flat_scan_wide_kernel     round 1's form (RR_WIDE_PD=0, A/B runs)
#include "flat_scan_common.h"

namespace rr {

// ---- wide rows (d > 1536, any multiple of 128), hand-pipelined -------------------------------------------------------
// The transposed design of flat_scan16_kernel: there the QUERIES stay in registers and the corpus streams past; here rows
// are too wide for that, so the ACCUMULATORS stay (all 256 AGPRs: 256 rows x 64 queries per wave) and both operands stream
// in 64-wide K steps: corpus slab (NT 32-row tiles x 128 B = 32 KB) HBM -> LDS by LDS-DMA through a 3-slot ring (two K
// steps ahead), the wave's 64 queries x 128 B from L2 straight into registers one K step ahead (scalar base + lane offset
// loads, double-buffered; every CU re-reads the query block once per 256 rows: 1 B of L2 traffic per corpus byte).
// Per K step and wave: 32 A fragments (ring of 8 ds_read_b128) x 4 MFMA 16x16x32, 8 DMA pieces, 8 query loads.
// The LDS image of a slab is NT copies of flat_scan16_kernel's 4 KB k-group block, so addressing is shared.
// vmcnt queue at the top of step s (oldest first): DMA(s) | q(s), DMA(s+1) -> wait vmcnt(NT): only DMA(s+1) may be out.
// Accumulators are HARD-WIRED AGPRs a[4i .. 4i+3]: 64 tied "+a" operands (256 registers) defeat hipcc's allocator (it
// shuttled them through scratch and v_accvgpr_mov at every loop edge).  Every statement that touches them names all 256 as
// clobbers, so the compiler keeps nothing of its own in AGPRs and sizes the kernel's register file for them.
#define RR_ALL_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
template <typename T> struct Mfma16Fixed;
#define RR_MFMA16F(NAME, MNEMONIC, FRAG)                                                                             \
  template <> struct Mfma16Fixed<NAME> {                                                                             \
    template <int R, bool FIRST>                                                                                     \
    static __device__ __forceinline__ void run(FRAG a, FRAG b) {                                                     \
      if (FIRST) asm volatile(MNEMONIC " a[%2:%3], %0, %1, 0" ::"v"(a), "v"(b), "n"(R), "n"(R + 3) : RR_ALL_AGPRS);  \
      else asm volatile(MNEMONIC " a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(R), "n"(R + 3) : RR_ALL_AGPRS); \
    }                                                                                                                \
  };
RR_MFMA16F(_Float16, "v_mfma_f32_16x16x32_f16", f16x8)
RR_MFMA16F(__bf16, "v_mfma_f32_16x16x32_bf16", bf16x8)
#undef RR_MFMA16F
template <int R>
__device__ __forceinline__ f32x4 read_acc_fixed() {
  f32x4 v;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3) : RR_ALL_AGPRS);
  return v;
}

template <typename F>
__device__ __forceinline__ void query_load_frag(F& dst, uint32_t lane_off, const void* sbase, int imm) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_off), "s"(sbase), "n"(imm) : "memory");
}

// (Carrying the fragment ring across K steps - 4 slots, the barrier moved to fragment 24 of the step before - measured 3.53 vs
// 3.66 TB/s: no gain, like the same experiment on flat_scan16_kernel; not kept.)
template <int N, typename F>
__device__ __forceinline__ void vm_wait_tied8(F& r0, F& r1, F& r2, F& r3, F& r4, F& r5, F& r6, F& r7) {
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "n"(N) : "memory");
}
// K rotation (rows wider than 1536 elements).  Every K step of a group reads a 128-byte column slab of 256 rows; with a row pitch
// of 4 or 8 KiB (d = 2048, 4096) those 256 addresses fall on a handful of HBM channels, and workgroups that walk the K steps in
// the same order keep hitting the same ones: a single query streamed 2M x 2048 at 0.70 of the HBM peak against 0.84 at d = 1792.
// So the 256-row group m (global: first tile / 8) starts its K loop at column (5 m) mod KG - rows and queries alike, the dot
// product only changes its summation order.  The order is a function of the ROW GROUP alone (chunk boundaries are multiples of 8
// tiles, capi.hip), so a row scores bit-identically in the sample launch, in whichever chunk launch it falls and in every
// wide-row kernel.  Measured: one query 0.70 -> 0.83 (d = 2048) and 0.71 -> 0.855 (d = 4096), 64 queries 0.66 -> 0.73 (d = 4096); 256
// queries unchanged (compute-bound).
#ifndef RR_WIDE_KROT
#define RR_WIDE_KROT 5   // 0: every group starts at column 0 (A/B)
#endif
__device__ __forceinline__ bool wide_rotates(int KG) { return RR_WIDE_KROT && KG > 24 && !(KG & 31); }
// rotation of the 256-row group whose first tile is `tile` (global: a function of the rows alone)
__device__ __forceinline__ int wide_rotation_of_tile(uint32_t tile, int KG) {
  // only where the row pitch is a multiple of 4 KiB (d = 2048, 4096, 6144, 8192): other pitches spread over the channels by
  // themselves (d = 1792: one query 0.845 without, 0.80 with; d = 3072 and 5120: no gain), and d <= 1536 is also served by the
  // half-resident kernel, which has no rotation.  Rows wider than 4096 rotate within a window of 64 columns: with the whole K range
  // in flight the 4 MB query block of d = 8192 no longer fits an XCD's L2 (256 queries -2.9 %); windowed: one query 0.79 -> 0.83
  // (d = 8192) and 0.77 -> 0.82 (d = 6144), 256 queries unchanged.
  if (!wide_rotates(KG)) return 0;
  return (int)(((tile >> 3) * (uint32_t)RR_WIDE_KROT) % (uint32_t)(KG > 64 ? 64 : KG));
}
// ... of launch group `grp` (groups of tiles_per_group launch ordinals).  Segmented chunk launches (rotating widths only, i.e. never
// FeB4RAG's or MedRAG's encoder groups): the run holding the group is looked up from the start of the launch's table each time -
// a few scalar loads per 256-row group - rather than with a third cursor, whose SGPRs the wide-row kernels do not have to spare.
__device__ __forceinline__ int wide_group_rotation(const ScanArgs& a, uint32_t grp, int tiles_per_group, int KG) {
  if (!wide_rotates(KG)) return 0;
  const uint32_t j = grp * (uint32_t)tiles_per_group;
  uint32_t tile = a.tile_first + j * a.tile_stride;
  if (a.ranges) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v;
    uint32_t r = 0;
    do {
      const RangeEntry* p = a.ranges + r;
      asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
      ++r;
    } while (j >= v[0]);   // (the last run's j_end is ~0)
    tile += v[1];
  }
  return wide_rotation_of_tile(tile, KG);
}
__device__ __forceinline__ int wide_rotated(int kg, int rot, int KG) { return kg + rot >= KG ? kg + rot - KG : kg + rot; }

template <typename F>
__device__ __forceinline__ void vm_drain_tied(F& r0, F& r1) {  // every vector-memory op has landed before r0 / r1 can be reused
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1)::"memory");
}
// DENSE launches (bootstrap sample, corpora <= 8192 rows) cover at most 256 tiles: with 8-tile groups only 32 workgroups would
// have work and each would still walk the whole K loop (112 us at d = 4096), so they use one tile per group.
template <typename T, bool DENSE, bool L2 = false>
__global__ __launch_bounds__(256, 1) void flat_scan_wide_kernel(const ScanArgs a, const int D) {
  typedef typename Mfma<T>::frag frag;
  constexpr int NT = DENSE ? 1 : 8;        // 32-row tiles per group
  constexpr int STEP_BYTES = NT * 4096;    // one K step of one group in LDS
  constexpr int NS = 3;
  constexpr int LEAD = NS - 1;             // K steps the DMA stream runs ahead
  constexpr int NF = 4 * NT;               // A fragments per K step: (tile, s2, rb)
  constexpr int NB = NF < 8 ? NF : 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, g = lane >> 4;
  const int KG = D / 64;                   // even (D is a multiple of 128)
  const uint32_t n_tiles = a.n_tiles;
  const uint32_t n_groups = (n_tiles + NT - 1) / NT;

  uint32_t roff[2];
  {
    const int rho = col & 7, p = col >> 3;
    const int f = ((rho >> 1) & 3) | (p << 2);
#pragma unroll
    for (int par = 0; par < 2; ++par) roff[par] = p * 1024 + rho * 128 + (((4 * par + g) ^ f) * 16);
  }
  // DMA side: this wave's piece of tile t = rows 8*wave .. +7, lane -> (row rho_w, 16-byte chunk c_w)
  const int rho_w = lane >> 3, sig = lane & 7;
  const int f_w = ((rho_w >> 1) & 3) | ((wave & 1) << 2);
  const int c_w = sig ^ f_w;
  const size_t row_bytes = (size_t)D * 2;
  int drot = 0;                            // K rotation of the DMA stream's group (wide_group_rotation)
  auto issue_piece = [&](uint32_t grp, int kg, int slot, int t) {
    uint32_t j = grp * NT + t;
    j = j < n_tiles ? j : n_tiles - 1;
    uint32_t row = (a.tile_first + j * a.tile_stride) * kTileRows + wave * 8 + rho_w;
    row = row < a.n_rows ? row : a.n_rows - 1;
    const char* gp = (const char*)a.xb + (size_t)row * row_bytes + (size_t)wide_rotated(kg, drot, KG) * 128 + c_w * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                     (__attribute__((address_space(3))) void*)(smem + slot * STEP_BYTES + t * 4096 + wave * 1024), 16, 0, 2);
  };

  const int nb = __builtin_amdgcn_readfirstlane(
      (int)a.nq <= wave * 64 ? 0 : ((int)a.nq - wave * 64 >= 64 ? 4 : ((int)a.nq - wave * 64 + 15) / 16));
  uint32_t qoff[4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const uint32_t qi = wave * 64 + qb * 16 + col;
    qoff[qb] = (qi < a.nq ? qi : a.nq - 1) * (uint32_t)(D * 2) + 16 * g;
  }
  LaneState4 st;
  lane_state_segments_init(a, st);
  const uint32_t nbuf = gridDim.x * 4;
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const uint32_t qi = wave * 64 + qb * 16 + col;
    st.thr[qb] = DENSE ? 0.f : a.thr[qi];
    st.cnt[qb] = 0;
    st.off[qb] = (qi * nbuf + blockIdx.x * 4 + g) * (uint32_t)a.cap;
  }

  uint32_t grp = blockIdx.x;               // group being multiplied
  uint32_t dgrp = blockIdx.x;              // group / K step the DMA stream is at (two steps ahead)
  int dkg = 0, dslot = 0;
  drot = wide_group_rotation(a, dgrp, NT, KG);       // K rotation of the DMA stream's group, of the compute group and of the next one
  int rot = drot, rot_next = wide_group_rotation(a, blockIdx.x + gridDim.x, NT, KG);
  auto dma_advance = [&]() {
    if (++dkg == KG) { dkg = 0; dgrp += gridDim.x; drot = wide_group_rotation(a, dgrp, NT, KG); }
    if (++dslot == NS) dslot = 0;
  };
  auto mfma4 = [&](auto r_tag, auto first_tag, frag x, frag q0, frag q1, frag q2, frag q3) {
    constexpr int R = decltype(r_tag)::value;
    constexpr bool F1 = decltype(first_tag)::value;
    Mfma16Fixed<T>::template run<R, F1>(x, q0);
    Mfma16Fixed<T>::template run<R + 4, F1>(x, q1);
    Mfma16Fixed<T>::template run<R + 8, F1>(x, q2);
    Mfma16Fixed<T>::template run<R + 12, F1>(x, q3);
  };
  auto read_tile = [&](auto r_tag, f32x4 (&e)[2][4]) {
    constexpr int R = decltype(r_tag)::value;
    e[0][0] = read_acc_fixed<R>();      e[0][1] = read_acc_fixed<R + 4>();  e[0][2] = read_acc_fixed<R + 8>();  e[0][3] = read_acc_fixed<R + 12>();
    e[1][0] = read_acc_fixed<R + 16>(); e[1][1] = read_acc_fixed<R + 20>(); e[1][2] = read_acc_fixed<R + 24>(); e[1][3] = read_acc_fixed<R + 28>();
  };
  frag q[2][4][2];                         // [buffer][query block][k slice]
  // The query loads are asynchronous asm: their destination registers must have LANDED before any point where hipcc may
  // copy them (it inserts v_mov copies of loop-carried values at loop back-edges).  So the wait for the queries of step
  // s+1 closes step s, and the statement names the registers as in/out operands: every later use or copy follows it.
  auto queries_landed = [&](auto buf_tag) {
    constexpr int B = decltype(buf_tag)::value;
    vm_wait_tied8<NT>(q[B][0][0], q[B][0][1], q[B][1][0], q[B][1][1], q[B][2][0], q[B][2][1], q[B][3][0], q[B][3][1]);
  };
  auto load_queries = [&](int buf, int kg) {
    const char* sb = (const char*)a.xq + (size_t)kg * 128;   // kg: the (rotated) column, chosen by the caller
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      query_load_frag(q[buf][qb][0], qoff[qb], sb, 0);
      query_load_frag(q[buf][qb][1], qoff[qb], sb, 64);
    }
  };
  if (grp < n_groups) {
    // prologue, in the steady-state queue order: DMA(0) [, DMA(1)] | q(0), DMA(LEAD-1)
#pragma unroll
    for (int sidx = 0; sidx < LEAD - 1; ++sidx) {
#pragma unroll
      for (int t = 0; t < NT; ++t) issue_piece(dgrp, dkg, dslot, t);
      dma_advance();
    }
    load_queries(0, wide_rotated(0, rot, KG));
#pragma unroll
    for (int t = 0; t < NT; ++t) issue_piece(dgrp, dkg, dslot, t);
    dma_advance();
    queries_landed(std::integral_constant<int, 0>{});   // only DMA(LEAD-1) may still be in flight
  }

  int slot = 0;   // accumulator of (tile t, row block rb, query block qb): a[16 (2t + rb) + 4 qb ..+3]
  while (grp < n_groups) {
    for (int kg = 0; kg < KG; kg += 2) {
      auto step = [&](auto buf_tag, auto first_tag, int kgs) {
        constexpr int P = decltype(buf_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        // (this step's queries and, older in the queue, its slab pieces were waited for when the previous step closed)
        __builtin_amdgcn_s_barrier();
        {
          const int nkg = kgs + 1 == KG ? wide_rotated(0, rot_next, KG) : wide_rotated(kgs + 1, rot, KG);   // the next group starts over on the same queries
          load_queries(1 - P, nkg);
        }
        if (nb > 0) {
          frag c[NB];
          const uint32_t ab0 = (uint32_t)(slot * STEP_BYTES) + roff[0], ab1 = (uint32_t)(slot * STEP_BYTES) + roff[1];
          // fragment f: tile f>>2, k slice (f>>1)&1, row block f&1
#pragma unroll
          for (int f = 0; f < NB; ++f) lds_read_frag(c[f], ((f >> 1) & 1) ? ab1 : ab0, (f >> 2) * 4096 + (f & 1) * 2048);
          static_for<NF>([&](auto fi) {
            constexpr int f = decltype(fi)::value;
            constexpr int t = f >> 2, par = (f >> 1) & 1, rb = f & 1;
            if (NF - f >= NB) lgkm_wait<NB - 1>();
            else if (NF - f == 7) lgkm_wait<6>();
            else if (NF - f == 6) lgkm_wait<5>();
            else if (NF - f == 5) lgkm_wait<4>();
            else if (NF - f == 4) lgkm_wait<3>();
            else if (NF - f == 3) lgkm_wait<2>();
            else if (NF - f == 2) lgkm_wait<1>();
            else lgkm_wait<0>();
            mfma4(std::integral_constant<int, 16 * (2 * t + rb)>{}, std::integral_constant<bool, FIRST && par == 0>{}, c[f % NB],
                  q[P][0][par], q[P][1][par], q[P][2][par], q[P][3][par]);
            if (f + NB < NF) {
              constexpr int fn = f + NB;
              lds_read_frag(c[f % NB], ((fn >> 1) & 1) ? ab1 : ab0, (fn >> 2) * 4096 + (fn & 1) * 2048);
            }
            if ((f & 3) == 1) issue_piece(dgrp, dkg, dslot, f >> 2);
          });
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t) issue_piece(dgrp, dkg, dslot, t);
        }
        dma_advance();
        if (++slot == NS) slot = 0;
        queries_landed(std::integral_constant<int, 1 - P>{});  // queue: ... q(s+1) | DMA(s+LEAD): NT pieces may remain
      };
      if (kg == 0) step(std::integral_constant<int, 0>{}, std::true_type{}, kg);
      else step(std::integral_constant<int, 0>{}, std::false_type{}, kg);
      step(std::integral_constant<int, 1>{}, std::false_type{}, kg + 1);
    }
    if (nb > 0) {
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      static_for<NT>([&](auto ti) {
        constexpr int t = decltype(ti)::value;
        const uint32_t j = grp * NT + t;
        if (j < n_tiles) {
          f32x4 e[2][4];
          read_tile(std::integral_constant<int, 32 * t>{}, e);
          if (L2) {  // rank by q.x - |x|^2/2: the lane's 2 x 4 rows of this tile (rows past the end are filtered by id later)
            const uint32_t r0 = (a.tile_first + j * a.tile_stride) * kTileRows + 4 * g;
            f32x4 h0, h1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              h0[i] = a.half_sqnorm[r0 + i < a.n_rows ? r0 + i : a.n_rows - 1];
              h1[i] = a.half_sqnorm[r0 + 16 + i < a.n_rows ? r0 + 16 + i : a.n_rows - 1];
            }
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
              e[0][qb] -= h0;
              e[1][qb] -= h1;
            }
          }
          tile_epilogue16<DENSE, 4>(a, st, e, j, lane, wave);
        }
      });
    }
    grp += gridDim.x;
    rot = rot_next;
    rot_next = wide_group_rotation(a, grp + gridDim.x, NT, KG);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
  if (!DENSE) {
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) a.cand_cnt[(wave * 64 + qb * 16 + col) * nbuf + blockIdx.x * 4 + g] = st.cnt[qb];
  }
}


// ---- deeper pipeline (round 2) ------------------------------------------------------------------------------------------
// PMC of the kernel above (profiles/r02/wide_*): HBM bytes = algorithmic (nothing re-read from HBM; the query re-stream hits
// L2), but the waves sit in s_waitcnt / s_barrier for 26 % of their cycles (the d <= 768 kernel: 14 %) and the matrix cores
// are busy 53 % (77 %).  Cause: vmcnt retires in order and the query loads of step s+1 are issued AFTER the DMA of steps
// s+1 .. s+LEAD-1, so waiting for those queries at the end of step s also waits for every slab issued before them: whatever
// the ring depth, only ONE slab (32 KB per CU) is ever in flight past a step boundary - 2/3 of what the d <= 768 kernel keeps.
// Here the queries are prefetched PD steps ahead into PD+1 register buffers and the slab ring runs PD+1 steps ahead
// (PD+2 slots), so the closing wait of step s - vmcnt(8 (2 PD - 1)) - leaves PD slabs and PD-1 query sets in flight:
//   issue order      ... DMA(s+1) | q(s+1) | DMA(s+2) | q(s+2) | ... | DMA(s+PD+1)
//   closing wait(s)      <- landed ------->| <- may still be in flight ----------->
// The query buffers are INPUT-ONLY operands of every asm statement that touches them (the loads included): hipcc sees PD+1
// sets of loop-invariant values that it must keep in fixed registers, and has no definition point inside the loop at which
// it could insert the register copies that read a destination before its load has landed (the failure mode of "=v" loads
// whose values are live around a back-edge).  wave_compact is inlined here: a call would spill the caller-saved part of
// those buffers around it and restore stale copies over loads that landed meanwhile.
// (the base goes through readfirstlane: it is uniform by construction, but should hipcc's divergence analysis ever think
// otherwise it would hand the "s" operand a VGPR pair - seen in a diagnostic build - and the instruction would not assemble)
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const void*)(((uint64_t)hi << 32) | lo);
}
template <typename F>
__device__ __forceinline__ void query_load_into(const F& dst, uint32_t lane_off, const void* sbase, int imm) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" ::"v"(dst), "v"(lane_off), "s"(uniform_ptr(sbase)), "n"(imm) : "memory");
}
template <typename T> struct Mfma16FixedIn;
#define RR_MFMA16FI(NAME, MNEMONIC, FRAG)                                                                            \
  template <> struct Mfma16FixedIn<NAME> {                                                                           \
    template <int R, bool FIRST>                                                                                     \
    static __device__ __forceinline__ void run(const FRAG& a, const FRAG& b) {                                       \
      if (FIRST) asm volatile(MNEMONIC " a[%2:%3], %0, %1, 0" ::"v"(a), "v"(b), "n"(R), "n"(R + 3) : RR_ALL_AGPRS);  \
      else asm volatile(MNEMONIC " a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(R), "n"(R + 3) : RR_ALL_AGPRS); \
    }                                                                                                                \
  };
RR_MFMA16FI(_Float16, "v_mfma_f32_16x16x32_f16", f16x8)
RR_MFMA16FI(__bf16, "v_mfma_f32_16x16x32_bf16", bf16x8)
#undef RR_MFMA16FI

#ifndef RR_WIDE_SPREAD
#define RR_WIDE_SPREAD 1   // measured at d = 4096, 2M rows, 256 queries: 0.496 vs 0.480 of the HBM peak (scan launches)
#endif
#if RR_WIDE_QFRAG
#define RR_WIDE_QBASE(a) ((const char*)(a).xqs)
#define RR_WIDE_QSTEP 2048     // bytes per 64-wide K step: two 32-wide k slices of 64 lanes x 16 B
#define RR_WIDE_QPAR 1024
#else
#define RR_WIDE_QBASE(a) ((const char*)(a).xq)
#define RR_WIDE_QSTEP 128
#define RR_WIDE_QPAR 64
#endif
#ifndef RR_WIDE_ABL
#define RR_WIDE_ABL 0   // development: timing-only ablations of the step (1 barrier, 2 query loads, 4 DMA, 8 LDS reads, 16 MFMA)
#endif
#if RR_WIDE_ABL && !defined(RR_DEV_VARIANTS)
#error "RR_WIDE_ABL removes work from the step (wrong scores, timing only): RR_DEV_VARIANTS builds only"
#endif
template <typename T, bool DENSE, bool L2, int NQB, int PD>
__global__ __launch_bounds__(256, 1) void flat_scan_wide_pd_kernel(const ScanArgs a, const int D) {
  typedef typename Mfma<T>::frag frag;
  constexpr int NT = DENSE ? 1 : 8;        // 32-row tiles per group
  constexpr int STEP_BYTES = NT * 4096;    // one K step of one group in LDS
  constexpr int QB = PD + 1;               // query register buffers
  constexpr int LEAD = PD + 1;             // K steps the DMA stream runs ahead
  constexpr int NS = LEAD + 1;             // LDS ring slots
  constexpr int NF = 4 * NT;               // A fragments per K step: (tile, s2, rb)
  constexpr int NB = NF < 8 ? NF : 8;
  constexpr int NQL = 2 * NQB;             // query loads per step and wave
  // The batch is dealt to the four waves in blocks of 16 queries, NQB consecutive blocks per wave, and the kernel is instantiated
  // for NQB = 1 ... 4 (up to 64 / 128 / 192 / 256 queries), so that no SIMD carries more than a quarter of the MFMAs.  (Consecutive,
  // not round robin: a wave computes all NQB slots of its instance or none, so 9 blocks are 3 + 3 + 3 + 0 = 9 computed slots
  // against 12 round robin - same busiest wave, fewer MFMAs under the power cap: 65 queries at d = 4096 0.81 against 0.74.)
  constexpr int QPW = 16 * NQB;
  constexpr int WAIT_Q = NT * PD + NQL * (PD - 1);  // ops younger than q(s+1) at the end of step s (waves holding queries)
  constexpr int WAIT_0 = NT * PD;                   // ... for a wave without queries: DMA(s+2) .. DMA(s+LEAD)
  static_assert(WAIT_Q <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, g = lane >> 4;
  const int KG = D / 64;                   // even (D is a multiple of 128), >= 14
  const uint32_t n_tiles = a.n_tiles;
  const uint32_t n_groups = (n_tiles + NT - 1) / NT;

  uint32_t roff[2];
  {
    const int rho = col & 7, p = col >> 3;
    const int f = ((rho >> 1) & 3) | (p << 2);
#pragma unroll
    for (int par = 0; par < 2; ++par) roff[par] = p * 1024 + rho * 128 + (((4 * par + g) ^ f) * 16);
  }
  // DMA side: this wave's piece of tile t = rows 8*wave .. +7, lane -> (row rho_w, 16-byte chunk c_w)
  const int rho_w = lane >> 3, sig = lane & 7;
  const int f_w = ((rho_w >> 1) & 3) | ((wave & 1) << 2);
  const int c_w = sig ^ f_w;
  // Source addresses: one uniform base per (group, K step) + a 32-bit lane offset per tile, recomputed once per group (the
  // per-piece 64-bit row arithmetic of the round-1 kernel cost 9 VALU + 5 SALU per piece, 8 pieces per step).
  const size_t row_bytes = (size_t)D * 2;
  const char* dbase = (const char*)a.xb;   // uniform: first row of the DMA stream's group, + K offset
  uint32_t voff[NT];
  int drot = 0;                            // K rotation of the DMA stream's group (wide_group_rotation)

  TileCursor dcur;                         // segmented search: cursor of the DMA stream (see TileCursor)
  cursor_init(a, dcur);
  auto dma_new_group = [&](uint32_t grp) {
    if (grp >= n_groups) return;           // the stream runs ahead of the last group: it re-reads that one (valid memory, never used)
    const uint32_t tile0 = cursor_tile(a, dcur, grp * NT);      // (a group never straddles two runs: runs are whole groups)
    const uint32_t row_base = tile0 * kTileRows;   // < n_rows: the group's first tile exists
    dbase = (const char*)a.xb + (size_t)row_base * row_bytes;
    drot = wide_rotation_of_tile(tile0, KG);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      uint32_t j = grp * NT + t;
      j = j < n_tiles ? j : n_tiles - 1;
      uint32_t row = (a.tile_first + j * a.tile_stride + (uint32_t)dcur.delta) * kTileRows + wave * 8 + rho_w;
      row = row < a.n_rows ? row : a.n_rows - 1;
      voff[t] = (row - row_base) * (uint32_t)row_bytes + c_w * 16;   // < 256 rows x 16 KB
    }
  };
  auto issue_piece = [&](int kg, int slot, int t) {
    const char* sb = dbase + (size_t)wide_rotated(kg, drot, KG) * 128;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + voff[t]),
                                     (__attribute__((address_space(3))) void*)(smem + slot * STEP_BYTES + t * 4096 + wave * 1024), 16, 0, 2);
  };

  // query blocks of this wave that hold real queries (wave-uniform); a kernel instance serves NQB blocks per wave
  const int nb = __builtin_amdgcn_readfirstlane((int)a.nq <= wave * QPW ? 0 : ((int)a.nq - wave * QPW + 15) / 16);
  uint32_t qoff[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
#if RR_WIDE_QFRAG   // [16-query block wave*NQB+qb][k slice][lane][8]: slots past nq repeat the last query (prep_kernel)
    qoff[qb] = (uint32_t)(wave * NQB + qb) * (uint32_t)(D * 32) + 16 * lane;
#else
    const uint32_t qi = wave * QPW + qb * 16 + col;
    qoff[qb] = (qi < a.nq ? qi : a.nq - 1) * (uint32_t)(D * 2) + 16 * g;
#endif
  }
  LaneState4 st;
  lane_state_segments_init(a, st);
  const uint32_t nbuf = gridDim.x * 4;
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const uint32_t qi = wave * QPW + (qb < NQB ? qb : 0) * 16 + col;   // (slots past NQB are never used)
    st.thr[qb] = (DENSE || qb >= NQB) ? 0.f : a.thr[qi];
#if RR_WIDE_ABL
    st.thr[qb] = __builtin_inff();   // ablated steps produce garbage scores: nothing may enter the insertion path
#endif
    st.cnt[qb] = 0;
    st.off[qb] = (qi * nbuf + blockIdx.x * 4 + g) * (uint32_t)a.cap;
  }
  // the thresholds are global loads: land them here, or their first use inside the loop drains the DMA ring
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(st.thr[0]), "+v"(st.thr[1]), "+v"(st.thr[2]), "+v"(st.thr[3]));

  frag q[QB][NQB][2];                      // [buffer][query block][k slice]: input-only operands everywhere (see above)
#pragma unroll
  for (int b = 0; b < QB; ++b)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      asm volatile("; query buffer %0" : "=v"(q[b][qb][0]));
      asm volatile("; query buffer %0" : "=v"(q[b][qb][1]));
    }
  auto load_queries = [&](auto buf_tag, int kg) {
    constexpr int B = decltype(buf_tag)::value;
    const char* sb = RR_WIDE_QBASE(a) + (size_t)kg * RR_WIDE_QSTEP;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      query_load_into(q[B][qb][0], qoff[qb], sb, 0);
      query_load_into(q[B][qb][1], qoff[qb], sb, RR_WIDE_QPAR);
    }
  };

  uint32_t grp = blockIdx.x;               // group being multiplied
  uint32_t dgrp = blockIdx.x;              // group / K step the DMA stream is at (LEAD steps ahead)
  int dkg = 0, dslot = 0;
  int rot = wide_group_rotation(a, grp, NT, KG), rot_next = wide_group_rotation(a, grp + gridDim.x, NT, KG);   // query side
  auto dma_advance = [&]() {
    if (++dkg == KG) { dkg = 0; dgrp += gridDim.x; dma_new_group(dgrp); }
    if (++dslot == NS) dslot = 0;
  };
  auto dma_slab = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) issue_piece(dkg, dslot, t);
    dma_advance();
  };
  auto closing_wait = [&]() {
#if (RR_WIDE_ABL & 6) == 6   // timing-only ablations (wrong results): count only what is still issued
#elif RR_WIDE_ABL & 2
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_0) : "memory");
#elif RR_WIDE_ABL & 4
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQL * (PD - 1)) : "memory");
#else
    if (nb > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_Q) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_0) : "memory");
#endif
  };
  if (grp < n_groups) {
    dma_new_group(dgrp);
    // prologue in the steady-state issue order: DMA(0) | q(0) DMA(1) | q(1) DMA(2) | ... | q(PD-1) DMA(PD)
    dma_slab();
    static_for<PD>([&](auto i) {
      if (nb > 0) load_queries(i, wide_rotated(decltype(i)::value, rot, KG));
      dma_slab();
    });
    closing_wait();                        // DMA(0) and q(0) have landed
  }

  int slot = 0, phase = 0, kg = 0;
  // accumulator of (tile t, row block rb, query block qb): a[16 (2t + rb) + 4 qb ..+3]
  auto step = [&](auto phase_tag, auto first_tag) {
    constexpr int P = decltype(phase_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int PN = (P + PD) % QB;      // buffer the queries of step s+PD go to (= the buffer of step s-1)
#if !(RR_WIDE_ABL & 1)
    __builtin_amdgcn_s_barrier();          // slab s is complete in LDS (every wave waited for its pieces), slab s-1 is free
#endif
    if (nb > 0) {
      // column of the queries PD steps ahead; past the end of this group the next one starts over on the same queries, in ITS order
      const int nkg = kg + PD >= KG ? wide_rotated(kg + PD - KG, rot_next, KG) : wide_rotated(kg + PD, rot, KG);
#if RR_WIDE_SPREAD
      // The CU's four waves run this code in lockstep (one barrier per step) and share ONE address unit: issued together, their
      // vector-memory instructions queue behind each other there and each holds its in-order wave ~60 cycles, with the matrix
      // pipe idle meanwhile.  So every wave starts its step 16 w cycles (one MFMA time) late, and the 16 VMEM instructions of a
      // step sit 2 fragments apart in the MFMA stream (queries in the first half, DMA in the second: same queue order).
      if (wave & 1) asm volatile("s_nop 15" ::: "memory");
      if (wave & 2) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      const char* qsb = RR_WIDE_QBASE(a) + (size_t)nkg * RR_WIDE_QSTEP;
#elif !(RR_WIDE_ABL & 2)
      load_queries(std::integral_constant<int, PN>{}, nkg);
#endif
      frag c[NB];
      const uint32_t ab0 = (uint32_t)(slot * STEP_BYTES) + roff[0], ab1 = (uint32_t)(slot * STEP_BYTES) + roff[1];
      // fragment f: tile f>>2, k slice (f>>1)&1, row block f&1
#pragma unroll
      for (int f = 0; f < NB; ++f) lds_read_frag(c[f], ((f >> 1) & 1) ? ab1 : ab0, (f >> 2) * 4096 + (f & 1) * 2048);
      static_for<NF>([&](auto fi) {
        constexpr int f = decltype(fi)::value;
        constexpr int t = f >> 2, par = (f >> 1) & 1, rb = f & 1;
        if (NF - f >= NB) lgkm_wait<NB - 1>();
        else if (NF - f == 7) lgkm_wait<6>();
        else if (NF - f == 6) lgkm_wait<5>();
        else if (NF - f == 5) lgkm_wait<4>();
        else if (NF - f == 4) lgkm_wait<3>();
        else if (NF - f == 3) lgkm_wait<2>();
        else if (NF - f == 2) lgkm_wait<1>();
        else lgkm_wait<0>();
#if !(RR_WIDE_ABL & 16)
        static_for<NQB>([&](auto qi) {
          constexpr int qb = decltype(qi)::value;
          Mfma16FixedIn<T>::template run<16 * (2 * t + rb) + 4 * qb, FIRST && par == 0>(c[f % NB], q[P][qb][par]);
        });
#endif
#if !(RR_WIDE_ABL & 8)
        if (f + NB < NF) {
          constexpr int fn = f + NB;
          lds_read_frag(c[f % NB], ((fn >> 1) & 1) ? ab1 : ab0, (fn >> 2) * 4096 + (fn & 1) * 2048);
        }
#endif
#if RR_WIDE_SPREAD
        if (NT == 8) {
#if !(RR_WIDE_ABL & 2)
          if constexpr ((f & 1) == 1 && f < 16 && (f >> 2) < NQB)
            query_load_into(q[PN][f >> 2][(f >> 1) & 1], qoff[f >> 2], qsb, ((f >> 1) & 1) * RR_WIDE_QPAR);
#endif
#if !(RR_WIDE_ABL & 4)
          if constexpr ((f & 1) == 1 && f >= 16) issue_piece(dkg, dslot, (f - 16) >> 1);
#endif
        } else {
          if (f == 0) load_queries(std::integral_constant<int, PN>{}, nkg);
          if ((f & 3) == 1) issue_piece(dkg, dslot, f >> 2);
        }
#elif !(RR_WIDE_ABL & 4)
        if ((f & 3) == 1) issue_piece(dkg, dslot, f >> 2);
#endif
      });
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) issue_piece(dkg, dslot, t);
    }
    dma_advance();
    if (++slot == NS) slot = 0;
    closing_wait();                        // slab s+1 (this wave's pieces) and q(s+1) have landed
  };
  auto read_tile = [&](auto r_tag, f32x4 (&e)[2][4]) {
    constexpr int R = decltype(r_tag)::value;
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) e[0][qb] = e[1][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_for<NQB>([&](auto qi) {
      constexpr int qb = decltype(qi)::value;
      e[0][qb] = read_acc_fixed<R + 4 * qb>();
      e[1][qb] = read_acc_fixed<R + 16 + 4 * qb>();
    });
  };
  while (grp < n_groups) {
    const bool first = kg == 0;
    static_for<QB>([&](auto pi) {
      if (phase == decltype(pi)::value) {
        if (first) step(pi, std::true_type{});
        else step(pi, std::false_type{});
      }
    });
    phase = phase + 1 == QB ? 0 : phase + 1;
    if (++kg < KG) continue;
    kg = 0;
    if (nb > 0) {
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      static_for<NT>([&](auto ti) {
        constexpr int t = decltype(ti)::value;
        const uint32_t j = grp * NT + t;
        if (j < n_tiles) {
          f32x4 e[2][4];
          read_tile(std::integral_constant<int, 32 * t>{}, e);
          if (L2) {  // rank by q.x - |x|^2/2: the lane's 2 x 4 rows of this tile (rows past the end are filtered by id later)
            const uint32_t r0 = (a.tile_first + j * a.tile_stride) * kTileRows + 4 * g;
            f32x4 h0, h1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              h0[i] = a.half_sqnorm[r0 + i < a.n_rows ? r0 + i : a.n_rows - 1];
              h1[i] = a.half_sqnorm[r0 + 16 + i < a.n_rows ? r0 + 16 + i : a.n_rows - 1];
            }
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
              e[0][qb] -= h0;
              e[1][qb] -= h1;
            }
          }
          tile_epilogue16<DENSE, NQB, QPW, true, 4>(a, st, e, j, lane, wave);
        }
      });
    }
    grp += gridDim.x;
    rot = rot_next;
    rot_next = wide_group_rotation(a, grp + gridDim.x, NT, KG);
  }
  // No LDS-DMA may outlive the workgroup - and no prefetched query load may outlive the LOOP: the sets of the steps that will
  // never run are still landing in the q registers, which hipcc regards as free from here on.  It computed the addresses of the
  // stores below in them BEFORE an untied wait (ALU work is not ordered by a "memory" clobber) and a late load then overwrote
  // the address: wild stores, intermittently, and only where the buffer in flight at loop exit was one the tail reuses (the
  // few-query kernel at d = 2048, KG mod 3 == 2).  The wait therefore names every q register as in/out: nothing of the tail
  // can be placed in them before it has executed.
  static_for<QB * NQB>([&](auto i) {
    constexpr int b = decltype(i)::value / NQB, qb = decltype(i)::value % NQB;
    vm_drain_tied(q[b][qb][0], q[b][qb][1]);
  });
  if (!DENSE) {
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) a.cand_cnt[(wave * QPW + qb * 16 + col) * nbuf + blockIdx.x * 4 + g] = st.cnt[qb];
  }
}


// ---- two waves per SIMD (round 2) ---------------------------------------------------------------------------------------
// Timing-only ablations of the kernel above (RR_WIDE_ABL; 2M x 4096, 256 queries, whole search): 4.30 ms as is, 3.70 without
// the query loads, 3.23 without the DMA instructions, 2.63 without both (LDS reads + MFMA + barrier alone), 4.12 without the
// barrier.  The 16 vector-memory instructions of a step cost ~40 % of it: each holds its in-order wave ~60 cycles at issue -
// whether or not the other waves issue theirs at the same moment (staggering them bought 0-3 %) - and with ONE wave per SIMD
// the matrix pipe idles meanwhile.  So: 512 threads, TWO waves per SIMD, each wave 256 rows x 32 queries (128 hard-wired
// AGPR accumulators, 128 VGPRs): while one wave of a SIMD is stuck issuing a load, its partner's MFMAs use the pipe.  Same
// LDS image, same ring and queue discipline as above with 4 DMA pieces and 4 query loads per wave and step; every A fragment
// is read from LDS by 8 waves instead of 4 (256 KB per step and CU = half of the LDS read rate).
#define RR_AGPRS_128 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
template <typename T> struct Mfma16Fixed128;
#define RR_MFMA16F128(NAME, MNEMONIC, FRAG)                                                                          \
  template <> struct Mfma16Fixed128<NAME> {                                                                          \
    template <int R, bool FIRST>                                                                                     \
    static __device__ __forceinline__ void run(const FRAG& a, const FRAG& b) {                                       \
      if (FIRST) asm volatile(MNEMONIC " a[%2:%3], %0, %1, 0" ::"v"(a), "v"(b), "n"(R), "n"(R + 3) : RR_AGPRS_128);  \
      else asm volatile(MNEMONIC " a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(R), "n"(R + 3) : RR_AGPRS_128); \
    }                                                                                                                \
  };
RR_MFMA16F128(_Float16, "v_mfma_f32_16x16x32_f16", f16x8)
RR_MFMA16F128(__bf16, "v_mfma_f32_16x16x32_bf16", bf16x8)
#undef RR_MFMA16F128
template <int R>
__device__ __forceinline__ f32x4 read_acc_fixed128() {
  f32x4 v;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3) : RR_AGPRS_128);
  return v;
}

#ifndef RR_WIDE8_SPREAD
#define RR_WIDE8_SPREAD 0
#endif
#ifndef RR_WIDE8_ABL
#define RR_WIDE8_ABL 0   // development, timing only (wrong results): 8 = every second LDS fragment read skipped, 256 = 8 extra reads, 32 = every query load
#endif                   // issued twice, 64 = nothing (baseline with the insertion path shut, as the others have it)
#if RR_WIDE8_ABL && !defined(RR_DEV_VARIANTS)
#error "RR_WIDE8_ABL removes work from the step (wrong scores, timing only): RR_DEV_VARIANTS builds only"
#endif
#ifndef RR_WIDE8_RING
#define RR_WIDE8_RING 8   // A fragments in flight from LDS per wave
#endif
template <typename T, bool L2, int NQB, int PD>
__global__ __launch_bounds__(512) void flat_scan_wide8_kernel(const ScanArgs a, const int D) {
  typedef typename Mfma<T>::frag frag;
  constexpr int NT = 8;                    // 32-row tiles per group
  constexpr int STEP_BYTES = NT * 4096;    // one K step of one group in LDS
  constexpr int QPW = 16 * NQB;            // queries per wave: blocks of 16 dealt to the waves, NQB each (1 up to 64 queries, 2 up to 256)
  constexpr int QB = PD + 1, LEAD = PD + 1, NS = LEAD + 1;
  constexpr int NF = 4 * NT;               // A fragments per K step: (tile, s2, rb)
  constexpr int NB = RR_WIDE8_RING;
  constexpr int NPW = 4;                   // DMA pieces per wave and slab (32 pieces over 8 waves)
  constexpr int NQL = 2 * NQB;             // query loads per step and wave
  constexpr int WAIT_Q = NPW * PD + NQL * (PD - 1);
  constexpr int WAIT_0 = NPW * PD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..7
  const int col = lane & 15, g = lane >> 4;
  const int KG = D / 64;
  const uint32_t n_tiles = a.n_tiles;
  const uint32_t n_groups = (n_tiles + NT - 1) / NT;

  uint32_t roff[2];
  {
    const int rho = col & 7, p = col >> 3;
    const int f = ((rho >> 1) & 3) | (p << 2);
#pragma unroll
    for (int par = 0; par < 2; ++par) roff[par] = p * 1024 + rho * 128 + (((4 * par + g) ^ f) * 16);
  }
  // DMA side: wave w fills row octet o = w & 3 (rows 8 o .. +7) of tiles 4 (w >> 2) .. +3; lane -> (row rho_w, 16-byte chunk c_w)
  const int oct = wave & 3, tbase = (wave >> 2) * 4;
  const int rho_w = lane >> 3, sig = lane & 7;
  const int f_w = ((rho_w >> 1) & 3) | ((oct & 1) << 2);
  const int c_w = sig ^ f_w;
  const size_t row_bytes = (size_t)D * 2;
  const char* dbase = (const char*)a.xb;
  uint32_t voff[NPW];
  int drot = 0;
  TileCursor dcur;                         // segmented search: cursor of the DMA stream
  cursor_init(a, dcur);
  auto dma_new_group = [&](uint32_t grp) {
    if (grp >= n_groups) return;
    const uint32_t tile0 = cursor_tile(a, dcur, grp * NT);
    const uint32_t row_base = tile0 * kTileRows;
    dbase = (const char*)a.xb + (size_t)row_base * row_bytes;
    drot = wide_rotation_of_tile(tile0, KG);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      uint32_t j = grp * NT + tbase + i;
      j = j < n_tiles ? j : n_tiles - 1;
      uint32_t row = (a.tile_first + j * a.tile_stride + (uint32_t)dcur.delta) * kTileRows + oct * 8 + rho_w;
      row = row < a.n_rows ? row : a.n_rows - 1;
      voff[i] = (row - row_base) * (uint32_t)row_bytes + c_w * 16;
    }
  };
  auto issue_piece = [&](int kg, int slot, int i) {
    const char* sb = dbase + (size_t)wide_rotated(kg, drot, KG) * 128;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + voff[i]),
                                     (__attribute__((address_space(3))) void*)(smem + slot * STEP_BYTES + (tbase + i) * 4096 + oct * 1024), 16, 0, 2);
  };

  const int nb = __builtin_amdgcn_readfirstlane((int)a.nq <= wave * QPW ? 0 : ((int)a.nq - wave * QPW + 15) / 16);
  uint32_t qoff[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
#if RR_WIDE_QFRAG
    qoff[qb] = (uint32_t)(wave * NQB + qb) * (uint32_t)(D * 32) + 16 * lane;
#else
    const uint32_t qi = wave * QPW + qb * 16 + col;
    qoff[qb] = (qi < a.nq ? qi : a.nq - 1) * (uint32_t)(D * 2) + 16 * g;
#endif
  }
  LaneState4 st;
  lane_state_segments_init(a, st);
  const uint32_t nbuf = gridDim.x * 4;
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const uint32_t qi = wave * QPW + (qb < NQB ? qb : 0) * 16 + col;
    st.thr[qb] = qb < NQB ? a.thr[qi] : 0.f;
#if RR_WIDE_ABL || RR_WIDE8_ABL
    st.thr[qb] = __builtin_inff();
#endif
    st.cnt[qb] = 0;
    st.off[qb] = (qi * nbuf + blockIdx.x * 4 + g) * (uint32_t)a.cap;
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(st.thr[0]), "+v"(st.thr[1]));

  frag q[QB][NQB][2];
#pragma unroll
  for (int b = 0; b < QB; ++b)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      asm volatile("; query buffer %0" : "=v"(q[b][qb][0]));
      asm volatile("; query buffer %0" : "=v"(q[b][qb][1]));
    }
  auto load_queries = [&](auto buf_tag, int kg) {
    constexpr int B = decltype(buf_tag)::value;
    const char* sb = RR_WIDE_QBASE(a) + (size_t)kg * RR_WIDE_QSTEP;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      query_load_into(q[B][qb][0], qoff[qb], sb, 0);
      query_load_into(q[B][qb][1], qoff[qb], sb, RR_WIDE_QPAR);
    }
  };

  uint32_t grp = blockIdx.x, dgrp = blockIdx.x;
  int dkg = 0, dslot = 0;
  int rot = wide_group_rotation(a, grp, NT, KG), rot_next = wide_group_rotation(a, grp + gridDim.x, NT, KG);
  auto dma_advance = [&]() {
    if (++dkg == KG) { dkg = 0; dgrp += gridDim.x; dma_new_group(dgrp); }
    if (++dslot == NS) dslot = 0;
  };
  auto dma_slab = [&]() {
#pragma unroll
    for (int i = 0; i < NPW; ++i) issue_piece(dkg, dslot, i);
    dma_advance();
  };
  auto closing_wait = [&]() {
    if (nb > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_Q) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_0) : "memory");
  };
  if (grp < n_groups) {
    dma_new_group(dgrp);
    dma_slab();
    static_for<PD>([&](auto i) {
      if (nb > 0) load_queries(i, wide_rotated(decltype(i)::value, rot, KG));
      dma_slab();
    });
    closing_wait();
  }

  int slot = 0, phase = 0, kg = 0;
  // accumulator of (tile t, row block rb, query block qb): a[8 (2t + rb) + 4 qb ..+3]
  auto step = [&](auto phase_tag, auto first_tag) {
    constexpr int P = decltype(phase_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int PN = (P + PD) % QB;
    __builtin_amdgcn_s_barrier();
    if (nb > 0) {
      const int nkg = kg + PD >= KG ? wide_rotated(kg + PD - KG, rot_next, KG) : wide_rotated(kg + PD, rot, KG);
#if RR_WIDE8_SPREAD
      const char* qsb = RR_WIDE_QBASE(a) + (size_t)nkg * RR_WIDE_QSTEP;   // query loads go into the first half of the MFMA stream, 4 fragments apart
#else
      load_queries(std::integral_constant<int, PN>{}, nkg);
#if RR_WIDE8_ABL & 32
      load_queries(std::integral_constant<int, PN>{}, nkg);
#endif
#endif
      frag c[NB];
      const uint32_t ab0 = (uint32_t)(slot * STEP_BYTES) + roff[0], ab1 = (uint32_t)(slot * STEP_BYTES) + roff[1];
#pragma unroll
      for (int f = 0; f < NB; ++f)
        if (!(RR_WIDE8_ABL & 8) || !(f & 1)) lds_read_frag(c[f], ((f >> 1) & 1) ? ab1 : ab0, (f >> 2) * 4096 + (f & 1) * 2048);
      static_for<NF>([&](auto fi) {
        constexpr int f = decltype(fi)::value;
        constexpr int t = f >> 2, par = (f >> 1) & 1, rb = f & 1;
        constexpr int LEFT = NF - f;       // fragments not yet consumed; min(LEFT, NB) - 1 reads may stay in flight
        lgkm_wait<(LEFT >= NB ? NB : LEFT) - 1>();
        static_for<NQB>([&](auto qi) {
          constexpr int qb = decltype(qi)::value;
          Mfma16Fixed128<T>::template run<8 * (2 * t + rb) + 4 * qb, FIRST && par == 0>(c[f % NB], q[P][qb][par]);
        });
        if constexpr (f + NB < NF && (!(RR_WIDE8_ABL & 8) || !(f & 1))) {
          constexpr int fn = f + NB;
          lds_read_frag(c[f % NB], ((fn >> 1) & 1) ? ab1 : ab0, (fn >> 2) * 4096 + (fn & 1) * 2048);
        }
        // (timing-only ablation 256, with 8: eight extra LDS fragment reads per step into the ring slots 8 leaves unused - together the
        // LDS load of a row-split layout that reads its query fragments from LDS: 16 A + 8 B reads per wave and step instead of 32)
        if constexpr ((RR_WIDE8_ABL & 256) != 0 && (f & 3) == 1) lds_read_frag(c[f % NB], ab0, (f >> 2) * 4096);
#if RR_WIDE8_SPREAD
        if constexpr ((f & 3) == 1 && f < 16 && (f >> 3) < NQB) query_load_into(q[PN][f >> 3][(f >> 2) & 1], qoff[f >> 3], qsb, ((f >> 2) & 1) * RR_WIDE_QPAR);
        if constexpr ((f & 3) == 1 && f >= 16) issue_piece(dkg, dslot, (f - 16) >> 2);
#else
        if constexpr ((f & 7) == 3) issue_piece(dkg, dslot, f >> 3);
#endif
      });
    } else {
#pragma unroll
      for (int i = 0; i < NPW; ++i) issue_piece(dkg, dslot, i);
    }
    dma_advance();
    if (++slot == NS) slot = 0;
    closing_wait();
  };
  while (grp < n_groups) {
    const bool first = kg == 0;
    static_for<QB>([&](auto pi) {
      if (phase == decltype(pi)::value) {
        if (first) step(pi, std::true_type{});
        else step(pi, std::false_type{});
      }
    });
    phase = phase + 1 == QB ? 0 : phase + 1;
    if (++kg < KG) continue;
    kg = 0;
    if (nb > 0 && !(RR_WIDE8_ABL & 128)) {   // (timing-only ablation 128: no epilogue at all - prices what overlapping it could gain)
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      static_for<NT>([&](auto ti) {
        constexpr int t = decltype(ti)::value;
        const uint32_t j = grp * NT + t;
        if (j < n_tiles) {
          f32x4 e[2][4];
#pragma unroll
          for (int qb = 0; qb < 4; ++qb) e[0][qb] = e[1][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
          static_for<NQB>([&](auto qi) {
            constexpr int qb = decltype(qi)::value;
            e[0][qb] = read_acc_fixed128<16 * t + 4 * qb>();
            e[1][qb] = read_acc_fixed128<16 * t + 8 + 4 * qb>();
          });
          if (L2) {
            const uint32_t r0 = (a.tile_first + j * a.tile_stride) * kTileRows + 4 * g;
            f32x4 h0, h1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              h0[i] = a.half_sqnorm[r0 + i < a.n_rows ? r0 + i : a.n_rows - 1];
              h1[i] = a.half_sqnorm[r0 + 16 + i < a.n_rows ? r0 + 16 + i : a.n_rows - 1];
            }
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
              e[0][qb] -= h0;
              e[1][qb] -= h1;
            }
          }
          tile_epilogue16<false, NQB, QPW, true, 8>(a, st, e, j, lane, wave);
        }
      });
    }
    grp += gridDim.x;
    rot = rot_next;
    rot_next = wide_group_rotation(a, grp + gridDim.x, NT, KG);
  }
  // (the wait is tied to every q register: see the end of flat_scan_wide_pd_kernel)
  static_for<QB * NQB>([&](auto i) {
    constexpr int b = decltype(i)::value / NQB, qb = decltype(i)::value % NQB;
    vm_drain_tied(q[b][qb][0], q[b][qb][1]);
  });
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) a.cand_cnt[(wave * QPW + qb * 16 + col) * nbuf + blockIdx.x * 4 + g] = st.cnt[qb];
}

// ---- row-split 8-wave kernel (round 3): 209 ... 256 queries at d > 768 ------------------------------------------------------------
// The 8-wave kernel above is LDS-bound: each wave multiplies 256 rows x 32 queries, so every A fragment feeds two MFMAs and the CU
// reads 256 KB of LDS per K step (2048 LDS cycles against 2048 MFMA cycles).  Here the eight waves are 2 row halves x 4 query
// quarters - a wave multiplies 128 rows x 64 queries, an A fragment feeds FOUR MFMAs - and the queries of a K step are DMA'd into
// LDS once per CU (a second slab next to the corpus slab: same 8-rows-x-128-bytes pieces, same source-side swizzle, the wave
// reads its 8 query fragments from there at the start of the step).  Per step and CU: 64 KB through the vector-memory path as
// before, 128 KB (A) + 64 KB (B) of LDS reads instead of 256 KB, no query registers in flight (no prefetch-register hazards).
// Timing-only ablations priced it at +5 ... 7 % (profiles/r03/wide8_rowsplit_pricing.json).  Rings: corpus 3 slots (two steps ahead:
// HBM latency), queries 2 slots (one step ahead: they come from L2) = all 160 KB of LDS.  Two waves share each query, so a
// workgroup owns 8 candidate buffers per query (row half x lane quarter).
// L2 = true (round 4): ranks by q.x - |x|^2/2 like the other kernels' L2 forms - the epilogue subtracts the rows' half squared norms
// (a.half_sqnorm) from the accumulators before the filter; an L2 search is never segmented.
template <typename T, bool L2 = false>
__global__ __launch_bounds__(512) void flat_scan_wide_rs_kernel(const ScanArgs a, const int D) {
  typedef typename Mfma<T>::frag frag;
  constexpr int NT = 8;                    // 32-row tiles per group (256 rows)
  constexpr int SLAB = NT * 4096;          // one K step of 256 rows (or of the 256 queries) in LDS
  constexpr int NSC = 3, NSQ = 2;          // corpus / query ring slots
  constexpr int QBASE = NSC * SLAB;
  constexpr int NF = 16;                   // A fragments per wave and step: (k slice, tile of the wave's row half, row block)
  constexpr int NB = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..7
  const int rh = wave & 1, qq = wave >> 1;                             // row half (tiles 4 rh .. +3), query quarter (queries 64 qq .. +63)
  const int col = lane & 15, g = lane >> 4;
  const int KG = D / 64;
  const uint32_t n_tiles = a.n_tiles;
  const uint32_t n_groups = (n_tiles + NT - 1) / NT;

  uint32_t roff[2];
  {
    const int rho = col & 7, p = col >> 3;
    const int f = ((rho >> 1) & 3) | (p << 2);
#pragma unroll
    for (int par = 0; par < 2; ++par) roff[par] = p * 1024 + rho * 128 + (((4 * par + g) ^ f) * 16);
  }
  // DMA side, both slabs: wave w fills row octet o = w & 3 (rows / queries 8 o .. +7) of tiles 4 (w >> 2) .. +3
  const int oct = wave & 3, tbase = (wave >> 2) * 4;
  const int rho_w = lane >> 3, sig = lane & 7;
  const int f_w = ((rho_w >> 1) & 3) | ((oct & 1) << 2);
  const int c_w = sig ^ f_w;
  const size_t row_bytes = (size_t)D * 2;
  const char* dbase = (const char*)a.xb;
  uint32_t voff[4], qvoff[4];
  int drot = 0;
  TileCursor dcur;
  cursor_init(a, dcur);
  auto dma_new_group = [&](uint32_t grp) {
    if (grp >= n_groups) return;
    const uint32_t tile0 = cursor_tile(a, dcur, grp * NT);
    const uint32_t row_base = tile0 * kTileRows;
    dbase = (const char*)a.xb + (size_t)row_base * row_bytes;
    drot = wide_rotation_of_tile(tile0, KG);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t j = grp * NT + tbase + i;
      j = j < n_tiles ? j : n_tiles - 1;
      uint32_t row = (a.tile_first + j * a.tile_stride + (uint32_t)dcur.delta) * kTileRows + oct * 8 + rho_w;
      row = row < a.n_rows ? row : a.n_rows - 1;
      voff[i] = (row - row_base) * (uint32_t)row_bytes + c_w * 16;
    }
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t qrow = (uint32_t)((tbase + i) * 32 + oct * 8 + rho_w);
    qrow = qrow < a.nq ? qrow : a.nq - 1;              // slots past nq repeat the last query (their thresholds are +inf)
    qvoff[i] = qrow * (uint32_t)row_bytes + c_w * 16;  // < 256 x 16 KB
  }
  auto issue_cpiece = [&](int kg, int slot, int i) {
    const char* sb = dbase + (size_t)wide_rotated(kg, drot, KG) * 128;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + voff[i]),
                                     (__attribute__((address_space(3))) void*)(smem + slot * SLAB + (tbase + i) * 4096 + oct * 1024), 16, 0, 2);
  };
  auto issue_qpiece = [&](int kcol, int slot, int i) {   // kcol: the (rotated) 64-wide column group; default cache policy: every group re-reads it
    const char* sb = (const char*)a.xq + (size_t)kcol * 128;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + qvoff[i]),
                                     (__attribute__((address_space(3))) void*)(smem + QBASE + slot * SLAB + (tbase + i) * 4096 + oct * 1024), 16, 0, 0);
  };

  LaneState4 st;
  lane_state_segments_init(a, st);
  const uint32_t nbuf = gridDim.x * 8;
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const uint32_t qi = qq * 64 + qb * 16 + col;
    st.thr[qb] = a.thr[qi];
    st.cnt[qb] = 0;
    st.off[qb] = (qi * nbuf + blockIdx.x * 8 + rh * 4 + g) * (uint32_t)a.cap;
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(st.thr[0]), "+v"(st.thr[1]), "+v"(st.thr[2]), "+v"(st.thr[3]));

  uint32_t grp = blockIdx.x, dgrp = blockIdx.x;
  int dkg = 0, dslot = 0;
  int rot = wide_group_rotation(a, grp, NT, KG), rot_next = wide_group_rotation(a, grp + gridDim.x, NT, KG);   // query side
  auto dma_advance = [&]() {
    if (++dkg == KG) { dkg = 0; dgrp += gridDim.x; dma_new_group(dgrp); }
    if (++dslot == NSC) dslot = 0;
  };
  // queue discipline (in-order vmcnt): per step a wave issues the 4 query pieces of step s+1 FIRST, then the 4 corpus pieces of
  // step s+2; "all but the 4 youngest have landed" at the end of step s therefore means: queries s+1 and corpus s+1 are in LDS
  auto closing_wait = [&]() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); };
  if (grp < n_groups) {
    dma_new_group(dgrp);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_cpiece(dkg, dslot, i);          // corpus step 0
    dma_advance();
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_qpiece(wide_rotated(0, rot, KG), 0, i);   // queries step 0
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_cpiece(dkg, dslot, i);          // corpus step 1
    dma_advance();
    closing_wait();
  }

  int cslot = 0, qslot = 0, kg = 0;
  // accumulator of (tile t of the row half, row block rb, query block qb): a[4 ((2 t + rb) 4 + qb) ..+3]
  auto step = [&](auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    __builtin_amdgcn_s_barrier();
    const int nkcol = kg + 1 >= KG ? wide_rotated(0, rot_next, KG) : wide_rotated(kg + 1, rot, KG);
    // (the wave's part of each slab - its query quarter = 2 "tiles" of 32 queries, its row half = 4 tiles - goes into the address
    // register, the rest is the instruction's immediate offset)
    uint32_t qa[2], ca[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      qa[par] = (uint32_t)(QBASE + qslot * SLAB + qq * 8192) + roff[par];
      ca[par] = (uint32_t)(cslot * SLAB + rh * 16384) + roff[par];
    }
    frag bq[2][4];            // B fragments of the step: [k slice][query block of the wave]
    frag c[NB];               // A fragment f lives in c[f % 8]; f: k slice par = f >> 3, tile t = (f >> 1) & 3 of the row half, row block rb = f & 1
    auto read_a = [&](auto fi) {
      constexpr int f = decltype(fi)::value;
      lds_read_frag(c[f % NB], ca[f >> 3], ((f >> 1) & 3) * 4096 + (f & 1) * 2048);
    };
    auto read_b = [&](auto pi, auto qi) {
      constexpr int par = decltype(pi)::value, qb = decltype(qi)::value;
      lds_read_frag(bq[par][qb], qa[par], (qb >> 1) * 4096 + (qb & 1) * 2048);
    };
    // Read schedule (LDS reads of a wave return in order, lgkmcnt(N) = all but the N youngest have landed).  The step starts right
    // after a barrier with EVERY wave of the CU reading, so the first MFMA must depend on as few reads as possible:
    //   issue order   A0 B00 B01 B02 B03 A1 .. A5 | after f=0: B10 A6 | f=1: B11 A7 | f=2: B12 A8 | f=3: B13 A9 | f=k (4..9): A(k+6)
    // (first version: all 8 B fragments + 8 A fragments up front - 72 reads CU-wide ahead of every wave's first MFMA, no gain over
    // the 8-wave kernel)
    read_a(std::integral_constant<int, 0>{});
    static_for<4>([&](auto qi) { read_b(std::integral_constant<int, 0>{}, qi); });
#ifndef RR_RS_AHEAD
#define RR_RS_AHEAD 4      // A fragments in flight ahead of the one being multiplied (4 or 6: same-device A/B 0.520-0.528 against 0.518-0.527)
#endif
    constexpr int AH = RR_RS_AHEAD;
    static_assert(AH == 4 || AH == 6, "RR_RS_AHEAD");
    static_for<AH - 1>([&](auto ai) { read_a(std::integral_constant<int, decltype(ai)::value + 1>{}); });
    static_for<NF>([&](auto fi) {
      constexpr int f = decltype(fi)::value;
      constexpr int par = f >> 3, t = (f >> 1) & 3, rb = f & 1;
      // reads issued before this fragment's MFMAs, and the issue index of A[f]:
      //   initial burst A0 B00..B03 A1..A(AH-1) = 4 + AH reads; after fragment f < 4: B1[f] and A(f+AH); after f >= 4: A(f+AH) while it exists
      constexpr int issued = f < 4 ? 4 + AH + 2 * f : f < NF - AH ? 8 + AH + f : 24;
      constexpr int idx_a = f == 0 ? 0 : f < AH ? 4 + f : f < AH + 4 ? 4 + AH + 2 * (f - AH) + 1 : 8 + f;
      constexpr int idx_b13 = 10 + AH;                             // the last B fragment of the second k slice
      constexpr int need = (par == 1 && idx_a < idx_b13) ? idx_b13 : idx_a;   // (with AH = 6, A8 is issued before B13)
      static_for<4>([&](auto qi) {
        constexpr int qb = decltype(qi)::value;
        if constexpr (f == 0) lgkm_wait<issued - 1 - (1 + qb)>();     // B0[qb] is read 1 + qb
        else if constexpr (qb == 0) lgkm_wait<issued - 1 - need>();    // (the B fragments of the fragment's k slice are covered by `need`)
        Mfma16Fixed128<T>::template run<4 * ((2 * t + rb) * 4 + qb), FIRST && par == 0>(c[f % NB], bq[par][qb]);
      });
      if constexpr (f < 4) read_b(std::integral_constant<int, 1>{}, std::integral_constant<int, f>{});
      if constexpr (f + AH < NF) read_a(std::integral_constant<int, f + AH>{});
      // the query pieces of step s+1 go out behind the first four fragments (they have only this one step to land: L2 latency; at
      // the very top of the step they would delay every wave's first LDS reads), the corpus pieces of step s+2 over the rest
      if constexpr (f < 4) issue_qpiece(nkcol, qslot ^ 1, f);
      if constexpr (f >= 5 && (f - 5) % 3 == 0) issue_cpiece(dkg, dslot, (f - 5) / 3);
    });
    dma_advance();
    if (++cslot == NSC) cslot = 0;
    qslot ^= 1;
    closing_wait();
  };
  while (grp < n_groups) {
    if (kg == 0) step(std::true_type{});
    else step(std::false_type{});
    if (++kg < KG) continue;
    kg = 0;
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    static_for<4>([&](auto ti) {
      constexpr int t = decltype(ti)::value;
      const uint32_t j = grp * NT + 4 * rh + t;
      if (j < n_tiles) {
        f32x4 e[2][4];
        static_for<4>([&](auto qi) {
          constexpr int qb = decltype(qi)::value;
          e[0][qb] = read_acc_fixed128<4 * ((2 * t) * 4 + qb)>();
          e[1][qb] = read_acc_fixed128<4 * ((2 * t + 1) * 4 + qb)>();
        });
        if (L2) {
          const uint32_t r0 = (a.tile_first + j * a.tile_stride) * kTileRows + 4 * g;
          f32x4 h0, h1;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            h0[i] = a.half_sqnorm[r0 + i < a.n_rows ? r0 + i : a.n_rows - 1];
            h1[i] = a.half_sqnorm[r0 + 16 + i < a.n_rows ? r0 + 16 + i : a.n_rows - 1];
          }
#pragma unroll
          for (int qb = 0; qb < 4; ++qb) {
            e[0][qb] -= h0;
            e[1][qb] -= h1;
          }
        }
        tile_epilogue16<false, 4, 64, true, 8>(a, st, e, j, lane, wave, qq);
      }
    });
    grp += gridDim.x;
    rot = rot_next;
    rot_next = wide_group_rotation(a, grp + gridDim.x, NT, KG);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) a.cand_cnt[(qq * 64 + qb * 16 + col) * nbuf + blockIdx.x * 8 + rh * 4 + g] = st.cnt[qb];
}

static int wide_pd() {  // RR_WIDE_PD: 0 = the round-1 kernel (one slab in flight), 2 / 3 = query prefetch distance of the deeper pipeline
  static const int v = [] {
    const char* e = tuning_env("RR_WIDE_PD");
    const int x = e ? atoi(e) : 2;
    return (x == 0 || x == 2 || x == 3) ? x : 2;
  }();
  return v;
}

// Filter launches: two waves per SIMD (flat_scan_wide8_kernel) up to d = 2048, one above.  Measured, 2M rows x 256 queries, whole
// search: d = 1024 3.53 vs 3.43 TB/s, 1536 3.61 vs 3.50, 2048 3.74 vs 3.64, 4096 3.74 vs 3.82.  RR_WIDE_WAVES=4|8 forces one.
static int wide_waves(int D) {
  static const int forced = [] {
    const char* e = tuning_env("RR_WIDE_WAVES");
    const int v = e ? atoi(e) : 0;
    return (v == 4 || v == 8) ? v : 0;
  }();
  return forced ? forced : (D <= 2048 ? 8 : 4);
}
// 129 ... 192 queries at d <= 2048: the 8-wave kernel deals 2 blocks per wave, so two SIMDs carry two busy waves and two carry one;
// the 4-wave kernel's 3-block instance is level.  Measured (2M x 2048 / 4M x 1024, scan fraction): 129 queries 0.61 -> 0.66,
// 160 queries 0.605 (8 waves) against 0.59, 192 queries 0.575 -> 0.60, 224 and 256 queries 8 waves (0.54 / 0.51 against 0.505 / 0.50).
static int wide_waves_for(int D, int nq) {
  const int w = wide_waves(D);
  static const bool forced = tuning_env("RR_WIDE_WAVES") != nullptr;
  if (forced || w != 8) return w;
  const int nblk = (nq + 15) / 16;
  return (nblk == 9 || nblk == 11 || nblk == 12) ? 4 : 8;
}

// 209 ... 256 queries at d > 768, 193 ... 256 at d > 2048 (round 4: either metric, any k; a workgroup owns 8 candidate buffers per query there): the row-split
// kernel (development builds: RR_WIDE_RS=0 / RR_WIDE_WAVES switch it off, RR_WIDE_RS_MAXK / _L2 restore round 3's limits for A/B runs).
// Measured against the kernels it replaces (scan launches, 256 queries): d = 1024 0.522-0.529 vs 0.510-0.515, 2048 0.527-0.530 vs
// 0.512-0.521, 3072 0.521-0.524 vs 0.510-0.512, 4096 0.525-0.526 vs 0.511-0.513, 8192 0.516-0.517 vs 0.501-0.505; with 208 queries
// or fewer the kernels that leave query quarters out win (160 queries: 0.61 against 0.52).
bool scan_wide_rowsplit(int D, int nq, bool l2, int k) {
  static const bool on = [] { const char* e = tuning_env("RR_WIDE_RS"); return !(e && atoi(e) == 0) && !tuning_env("RR_WIDE_WAVES"); }();
  static const int max_d = [] { const char* e = tuning_env("RR_WIDE_RS_MAXD"); return e ? atoi(e) : kMaxDim; }();   // (tuning runs)
  // fewest queries it takes: 209 at d <= 2048 (13 blocks run as well or better on the 8-wave kernel: 200 queries 0.524 / 0.531 against
  // 0.525-0.530 / 0.527-0.537 at d = 1024 / 2048, and its compaction has half the buffers), 193 above 2048, where the alternative is
  // the 4-wave kernel (193 / 200 / 208 queries at d = 4096: 0.530-0.534 against 0.509-0.512; profiles/r04/rowsplit_widening.json)
  static const int min_q_env = [] { const char* e = tuning_env("RR_WIDE_RS_MINQ"); return e ? atoi(e) : 0; }();
  const int min_q = min_q_env ? min_q_env : (D > 2048 ? 193 : 209);
  static const int max_k = [] { const char* e = tuning_env("RR_WIDE_RS_MAXK"); return e ? atoi(e) : kMaxK; }();
  static const bool with_l2 = [] { const char* e = tuning_env("RR_WIDE_RS_L2"); return !(e && atoi(e) == 0); }();
  return on && wide_pd() != 0 && (!l2 || with_l2) && D > kMaxResidentDim && D <= max_d && nq >= min_q && k <= max_k;
}
// name of the filter-launch kernel launch_scan_wide_t picks (reported by rr_flat_scan_kernel_name)
const char* scan_wide_kernel_name(int D, int nq) {
  if (scan_wide_rowsplit(D, nq, false, 32)) return "flat_scan_wide_rs_kernel";
  if (wide_pd() == 0) return "flat_scan_wide_kernel";
  return wide_waves_for(D, nq) == 8 ? "flat_scan_wide8_kernel" : "flat_scan_wide_pd_kernel";
}

template <typename T>
static hipError_t launch_scan_wide_t(const ScanArgs& a, int D, bool dense, int grid, hipStream_t st) {
  if (D % 128 != 0) return hipErrorInvalidValue;
  const int pd = wide_pd();
  const size_t lds = (size_t)(pd == 0 ? 3 : pd + 2) * (dense ? 1 : 8) * 4096;
  const bool l2 = a.half_sqnorm != nullptr;
  // query blocks (16 queries) per wave: the batch is dealt evenly to the waves, so 17 ... 128 queries no longer sit on one or two SIMDs
  const int nqb4 = a.nq <= 64 ? 1 : a.nq <= 128 ? 2 : a.nq <= 192 ? 3 : 4;   // four-wave kernel (consecutive blocks per wave)
  const bool one8 = a.nq <= 64;                            // eight-wave kernel: 1 block per wave up to 64 queries (measured at d = 2048,
                                                           // 128 queries: 2 blocks on four waves 0.76, 1 block on eight 0.73 - every LDS
                                                           // fragment read then feeds one MFMA instead of two)
  hipError_t e;
#define RR_LAUNCH_KERNEL(...)                                                                                        \
  {                                                                                                                  \
    e = hipFuncSetAttribute((const void*)__VA_ARGS__, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
    if (e != hipSuccess) return e;                                                                                   \
    hipLaunchKernelGGL((__VA_ARGS__), dim3(grid), dim3(256), lds, st, a, D);                                        \
    return hipGetLastError();                                                                                        \
  }
#ifdef RR_WIDE_PD3   // development builds only: the PD = 3 instantiations double the compile time of this unit
#define RR_PD3_CASE(DENSE_, L2_, NQB_) if (pd == 3) RR_LAUNCH_KERNEL(flat_scan_wide_pd_kernel<T, DENSE_, L2_, NQB_, 3>)
#else
#define RR_PD3_CASE(DENSE_, L2_, NQB_)
#endif
#define RR_LAUNCH_PD(DENSE_, L2_, NQB_)                                                   \
  {                                                                                       \
    RR_PD3_CASE(DENSE_, L2_, NQB_)                                                        \
    RR_LAUNCH_KERNEL(flat_scan_wide_pd_kernel<T, DENSE_, L2_, NQB_, 2>)                   \
  }
  if (!dense && scan_wide_rowsplit(D, (int)a.nq, l2, a.k)) {
    const size_t lds_rs = (size_t)5 * 8 * 4096;   // 3 corpus + 2 query slots of 32 KB: all 160 KB
    if (l2 && a.ranges) return hipErrorNotSupported;
#define RR_LAUNCH_RS(...)                                                                                            \
  {                                                                                                                  \
    hipError_t ers = hipFuncSetAttribute((const void*)__VA_ARGS__, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rs); \
    if (ers != hipSuccess) return ers;                                                                               \
    hipLaunchKernelGGL((__VA_ARGS__), dim3(grid), dim3(512), lds_rs, st, a, D);                                     \
    return hipGetLastError();                                                                                        \
  }
    if (l2) RR_LAUNCH_RS(flat_scan_wide_rs_kernel<T, true>)
    RR_LAUNCH_RS(flat_scan_wide_rs_kernel<T, false>)
#undef RR_LAUNCH_RS
  }
  if (!dense && pd != 0 && wide_waves_for(D, (int)a.nq) == 8) {
    hipError_t e8;
#define RR_LAUNCH_8(...)                                                                                             \
  {                                                                                                                  \
    e8 = hipFuncSetAttribute((const void*)__VA_ARGS__, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
    if (e8 != hipSuccess) return e8;                                                                                 \
    hipLaunchKernelGGL((__VA_ARGS__), dim3(grid), dim3(512), lds, st, a, D);                                        \
    return hipGetLastError();                                                                                        \
  }
    if (l2) { if (one8) RR_LAUNCH_8(flat_scan_wide8_kernel<T, true, 1, 2>) else RR_LAUNCH_8(flat_scan_wide8_kernel<T, true, 2, 2>) }
    if (one8) RR_LAUNCH_8(flat_scan_wide8_kernel<T, false, 1, 2>) else RR_LAUNCH_8(flat_scan_wide8_kernel<T, false, 2, 2>)
#undef RR_LAUNCH_8
  }
  if (pd == 0) {
    if (a.ranges) return hipErrorNotSupported;   // RR_WIDE_PD=0 (A/B of the round-1 kernel): plain searches only
    if (l2) { if (dense) RR_LAUNCH_KERNEL(flat_scan_wide_kernel<T, true, true>) else RR_LAUNCH_KERNEL(flat_scan_wide_kernel<T, false, true>) }
    if (dense) RR_LAUNCH_KERNEL(flat_scan_wide_kernel<T, true, false>) else RR_LAUNCH_KERNEL(flat_scan_wide_kernel<T, false, false>)
  }
#define RR_LAUNCH_PD_N(DENSE_, L2_)                      \
  {                                                      \
    if (nqb4 == 1) RR_LAUNCH_PD(DENSE_, L2_, 1)          \
    if (nqb4 == 2) RR_LAUNCH_PD(DENSE_, L2_, 2)          \
    if (nqb4 == 3) RR_LAUNCH_PD(DENSE_, L2_, 3)          \
    RR_LAUNCH_PD(DENSE_, L2_, 4)                         \
  }
  if (l2) {
    if (dense) RR_LAUNCH_PD_N(true, true)
    RR_LAUNCH_PD_N(false, true)
  }
  if (dense) RR_LAUNCH_PD_N(true, false)
  RR_LAUNCH_PD_N(false, false)
#undef RR_LAUNCH_PD_N
#undef RR_LAUNCH_PD
#undef RR_LAUNCH_KERNEL
}

hipError_t launch_scan_wide(const ScanArgs& a, int dtype, int D, bool dense, int grid, hipStream_t st) {
  return dtype == RR_DTYPE_F16 ? launch_scan_wide_t<_Float16>(a, D, dense, grid, st) : launch_scan_wide_t<__bf16>(a, D, dense, grid, st);
}

}  // namespace rr
