// bf16 GEMM for gfx950:  C[m][n] = sum_k A[m][k] * B[n][k]   (A, B bf16 with k contiguous; fp32 accumulate on
// v_mfma_f32_32x32x16_bf16; fused bias / residual / erf-GELU epilogues).  The Linear layers of Attention and
// FeedForward (GoalFormer.py:42-50,64,66-69) in the bf16 configuration (BASELINE config 5).
//
// Common to both kernels below (one workgroup per BM x BN output tile):
//  * Staging is LDS-DMA: `buffer_load_dwordx4 ... lds` writes 16 bytes per lane straight into LDS (no VGPRs, no
//    ds_write).  An LDS-DMA writes lane-linearly, so the bank swizzle of the LDS image is applied to the per-lane
//    SOURCE address and again by the fragment reads; every ds_read_b128 lane group {0-3,12-15,20-27} /
//    {4-11,16-19,28-31} then touches 16 distinct 16-byte bank slots (conflict-free).
//  * Rows beyond M / N and the k tail are zero-filled by the buffer range check (offset >= num_records reads 0),
//    so the main loops are branch-free.
//  * Each wave owns a (BM/WM) x (BN/WN) block of 32x32 accumulators; an A/B fragment is one ds_read_b128
//    (lane (i, h): row i, k = 16 s + 8 h .. + 7).
//  * Epilogue: accumulators (column on the lane, rows in registers) are transposed through a wave-private slice of
//    the now idle staging LDS so that global accesses are 16-byte (fp32) / 8-byte (bf16) row-contiguous pieces.
//
// gemm_bf16_ring_kernel (8 waves, 256-row tiles; the large GEMMs): the k-tile is 32 deep (64-byte rows, chunks swizzled
// by ((row >> 2) & 3)) and the LDS is a ring of NS such tiles with NS - 1 of them in flight.  Waves 0-3 and 4-7 (SIMD
// partners) run the same program one barrier interval apart ("ping-pong"): per k-tile a wave LOADS its fragments and
// issues the DMA of tile t + NS - 1, passes a barrier, COMPUTES 2 x MT x NTL MFMAs at raised priority, passes a
// barrier; every interval has one wave of each SIMD in its MFMA segment and its partner in its load segment.
// Measured: 1.4-1.7k cycles per k-tile against 512 MFMA-bound -- the load segment (12 fragment reads,
// 4 DMA issues, two barriers) and the L2 -> LDS delivery of 32 KB per k-tile (21-23 B/clk/CU) set the pace, not MFMA issue.
// gemm_bf16_kernel (4 waves, 128x128, BK = 64, two LDS buffers, one barrier per k-tile): small problems.
#include "bf16.h"
#include "kernels.h"


namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ int xcd_chunk(int id, int n) {
  // blocks are dealt round-robin over the 8 XCDs: give each XCD one contiguous chunk of the tile grid (bijective)
  const int q = n >> 3, r = n & 7, xcd = id & 7, loc = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

template <int BM_, int BN_, int WM_, int WN_>
struct BTile {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NW = WM_ * WN_, NT = 64 * NW;
  static constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NTL = WTN / 32;
  static constexpr int BUF = (BM + BN) * 128;            // bytes of one k-tile of A and B (BK = 64 bf16 = 128 B per row)
  static constexpr int EPW = 32 * WTN * 4;               // epilogue staging bytes per wave (32 rows x WTN fp32)
  static constexpr int LDS = 2 * BUF > NW * EPW ? 2 * BUF : NW * EPW;
  static constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;   // LDS-DMA instructions per wave and k-tile
  static_assert(WTM % 32 == 0 && WTN % 32 == 0 && NW % 2 == 0 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile shape");
};

// Transposes the wave's accumulators through its private LDS slice and applies the epilogue.
template <class T, int EPI>
__device__ __forceinline__ void epilogue(const GemmBf16Params& p, unsigned char* smem, f32x16 (&acc)[T::MT][T::NTL], int m0, int n0,
                                         int wave, int lane) {
  constexpr int MT = T::MT, NTL = T::NTL, WTM = T::WTM, WTN = T::WTN;
  const int li = lane & 31, h = lane >> 5;
  const int wr = wave / T::WN, wc = wave % T::WN;
  float* es = reinterpret_cast<float*>(smem + wave * T::EPW);
  constexpr int LPR = WTN / 4;        // lanes per output row piece
  constexpr int RPI = 64 / LPR;       // rows per read instruction
  const int erow = lane / LPR, ecol = (lane % LPR) * 4;
  const int gn = n0 + wc * WTN + ecol;
  const bool ncol = gn < p.N;
  fx4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && ncol) bias4 = *reinterpret_cast<const fx4*>(p.bias + gn);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) es[((r & 3) + 8 * (r >> 2) + 4 * h) * WTN + j * 32 + li] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 32 / RPI; ++q) {
      const int lr = q * RPI + erow;
      fx4 v = *reinterpret_cast<const fx4*>(es + lr * WTN + ecol);
      const int gm = m0 + wr * WTM + i * 32 + lr;
      if (gm < p.M && ncol) {
        v += bias4;
        const long long crow = p.c_rgrp > 0 ? gm + gm / p.c_rgrp + 1 : gm;
        if (EPI == BEPI_F32 || EPI == BEPI_F32_PLAIN) {
          if (EPI == BEPI_F32 && p.res) {
            const long long rr = p.res_mod > 0 ? (gm % p.res_mod) + 1 : gm;
            v += *reinterpret_cast<const fx4*>(p.res + rr * p.ldr + gn);
          }
          *reinterpret_cast<fx4*>(reinterpret_cast<float*>(p.C) + crow * p.ldc + gn) = v;
        } else if (EPI == BEPI_BF16) {
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + crow * p.ldc + gn) = __builtin_convertvector(v, bf16x4);
        } else if (EPI == BEPI_GELU_BF16 || EPI == BEPI_GELU2_BF16) {
          if (EPI == BEPI_GELU2_BF16) *reinterpret_cast<bf16x4*>(p.C2 + (long long)gm * p.ldc2 + gn) = __builtin_convertvector(v, bf16x4);
          fx4 g = gelu_bf16x4(v);
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + crow * p.ldc + gn) = __builtin_convertvector(g, bf16x4);
        } else {  // BEPI_DGELU_BF16
          const bf16x4 a = *reinterpret_cast<const bf16x4*>(p.aux + (long long)gm * p.ldaux + gn);
          fx4 g = {v[0] * gelu_erf_grad((float)a[0]), v[1] * gelu_erf_grad((float)a[1]), v[2] * gelu_erf_grad((float)a[2]),
                   v[3] * gelu_erf_grad((float)a[3])};
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + crow * p.ldc + gn) = __builtin_convertvector(g, bf16x4);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ simple kernel
// BK = 64 (128-byte rows, chunk swizzle ((row >> 1) & 7)), two LDS buffers, one barrier per k-tile.
template <class T, int EPI>
__global__ void __launch_bounds__(T::NT) gemm_bf16_kernel(const GemmBf16Params p) {
  constexpr int BM = T::BM, BN = T::BN, NW = T::NW, MT = T::MT, NTL = T::NTL, WTM = T::WTM, WTN = T::WTN;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_chunk(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int wr = wave / T::WN, wc = wave % T::WN;

  long long abytes = ((long long)(p.M - 1 - m0) * p.lda + p.K) * 2, bbytes = ((long long)(p.N - 1 - n0) * p.ldb + p.K) * 2;
  if (abytes > 0x7FFFFFF0ll) abytes = 0x7FFFFFF0ll;
  if (bbytes > 0x7FFFFFF0ll) bbytes = 0x7FFFFFF0ll;
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A + (long long)m0 * p.lda), 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B + (long long)n0 * p.ldb), 0, (int)bbytes, 0x00020000);
  const int srow = lane >> 3;                                    // row inside the 8-row group one instruction fills
  const int sc = (lane & 7) ^ (((wave & 1) * 4 + (srow >> 1)) & 7);   // logical 16-byte chunk this lane fetches
  const unsigned offA = ((unsigned)(wave * 8 + srow) * (unsigned)p.lda + sc * 8u) * 2u;
  const unsigned offB = ((unsigned)(wave * 8 + srow) * (unsigned)p.ldb + sc * 8u) * 2u;
  const unsigned stepA = (unsigned)(NW * 8) * (unsigned)p.lda * 2u, stepB = (unsigned)(NW * 8) * (unsigned)p.ldb * 2u;
  const int nkt = (p.K + 63) / 64;

  auto issue = [&](int t, int buf) {
    // chunks at k >= K are sent out of range (bit 31 set: beyond any num_records) and read as zero
    const unsigned dead = (t * 64 + sc * 8 < p.K) ? 0u : 0x80000000u;
    const unsigned oa = (offA + (unsigned)t * 128u) | dead, ob = (offB + (unsigned)t * 128u) | dead;
    unsigned char* dst = smem + buf * T::BUF + wave * 1024;
#pragma unroll
    for (int i = 0; i < T::GA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + i * NW * 1024), 16, oa + i * stepA, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < T::GB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(dst + BM * 128 + i * NW * 1024), 16, ob + i * stepB, 0, 0, 0);
  };

  // fragment addresses (bytes inside a buffer): row * 128 + ((2 s + h) ^ f(row)) * 16
  const unsigned fsw = (unsigned)((li >> 1) & 7);
  const unsigned a_l0 = (unsigned)(wr * WTM + li) * 128u + ((h ^ fsw) * 16u);
  const unsigned b_l0 = (unsigned)(BM + wc * WTN + li) * 128u + ((h ^ fsw) * 16u);

  f32x16 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nkt; ++t) {
    if (t + 1 < nkt) issue(t + 1, (t + 1) & 1);
    const unsigned char* sb = smem + (t & 1) * T::BUF;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 af[MT], bf[NTL];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + (a_l0 ^ (s * 32u)) + i * 4096);
#pragma unroll
      for (int j = 0; j < NTL; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(sb + (b_l0 ^ (s * 32u)) + j * 4096);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  epilogue<T, EPI>(p, smem, acc, m0, n0, wave, lane);
}

// ------------------------------------------------------------------------------------------------ ring kernel
// Persistent: one 8-wave workgroup per CU walks tiles blockIdx.x, + gridDim.x, ...; BK = 32: a k-tile is (BM + BN)
// rows x 64 bytes.  The k-tiles of ALL the workgroup's tiles form one continuous stream through a ring of NS LDS
// slots with D = NS - 1 tiles of LDS-DMA in flight: the loads run D k-tiles ahead of the MFMAs straight across tile
// boundaries, so a tile's prologue (first loads) hides under the previous tile's last MFMAs and its epilogue stores
// drain under the next tile's MFMAs.  Measured before this structure (one workgroup per tile, K = 768): prologue
// 7.3k + epilogue 12k cycles around a 34k-cycle main loop (tools/bf16_stamps.py).
// Waves 0-3 and 4-7 (SIMD partners) run the same program one barrier interval apart ("ping-pong"):
//   L(g): ds_read the fragments of stream k-tile g; issue the DMA of k-tile g + D into the slot k-tile g - 1 had;
//         wait until this wave's share of k-tile g + 1 has landed; wait lgkmcnt(0); barrier.
//   C(g): 2 x MT x NTL MFMAs at raised priority; barrier.   After the last C of a tile: epilogue (no barrier inside).
//   WAR: both groups finished L(g-1) (reads retired by lgkmcnt(0)) before a barrier that precedes any L(g).
//   RAW: a wave's share of k-tile g + 1 is waited for at the end of its L(g), before the barrier that precedes the
//        other group's and its own L(g + 1).
//   vmcnt: loads, LDS-DMAs and stores retire in issue order on one counter.  The epilogue's NST stores are issued
//        AFTER the DMAs of the next D - 1 k-tiles, so for the D - 1 load segments that follow an epilogue the wait
//        leaves NST more operations outstanding (the stores need not have drained); every store / residual load of the
//        epilogue is an unconditional buffer instruction (invalid rows / columns go out of range) so NST is exact.
template <class T, int EPI>
struct Ring {
  static constexpr int SLOT = (T::BM + T::BN) * 64;                 // bytes per k-tile
  static constexpr int EPW = 16 * T::WTN * 4;                       // epilogue staging per wave: 16 rows x WTN fp32
  static constexpr int NS = ((163840 - 8 * EPW) / SLOT) < 6 ? ((163840 - 8 * EPW) / SLOT) : 6;
  static constexpr int D = NS - 1;
  static constexpr int GA = T::BM / 16 / 8, GB = T::BN / 16 / 8;    // DMA instructions per wave and k-tile (16 rows each)
  static constexpr int PER = GA + GB;
  static constexpr int LDS = NS * SLOT + 8 * EPW;
  static constexpr int NST = T::MT * 2 * (16 * T::WTN / 256) * (EPI == BEPI_GELU2_BF16 ? 2 : 1);   // stores per wave and tile
  static constexpr int WAIT = PER * (D - 1);
  static constexpr int WAIT_EPI = WAIT + NST <= 63 ? WAIT + NST : WAIT;   // a smaller count only over-waits
  static_assert(T::NW == 8 && T::BM % 128 == 0 && T::BN % 128 == 0 && NS >= 3, "ring kernel: 8 waves, 128-row multiples");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
#define DGVIT_BUF_OOB 0x80000000u

// STAMP: diagnostic build (tools/bf16_stamps.py): wave 0 / wave 4 lane 0 of every workgroup write s_memtime at the start
// of each of its first 8 tiles, after the tile's main loop and after its epilogue, to a buffer nothing else reads.
// The shipped instantiations have STAMP = false (no stamp executes).
// TN: both operands are stored contraction-major -- A (K x M, row stride lda), B (K x N): the weight gradient
// dW = dY^T X straight from the token-major activations, no transposed copies.  A k-tile is then 32 rows x 512 bytes per
// operand (one DMA instruction = 2 rows; 16-byte chunk c of row r stored at chunk c ^ ((r & 3) << 2)), and a fragment is
// two `ds_read_b64_tr_b16`: per 16-lane group the hardware reads a 4 (k) x 16 (m) block and hands lane i column i, so a
// lane gets 4 consecutive k of its own m -- exactly the MFMA operand order.  The swizzle puts the block's 4 rows on the
// four 64-byte bank groups: a 32-lane half (two blocks) touches every bank once.
// M16: v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (same cycles per FLOP, same LDS bytes; the kernel runs under the chip's
// power limit on random data -- all-zero operands are 11-44 % faster -- and the 16x16 shape draws less: MI355X_MICROARCH.md,
// 'DVFS give-back' item 7).  A k-tile is exactly one 32-deep MFMA step: lane l holds row l & 15, k = 8 (l >> 4) .. + 7, i.e.
// chunk l >> 4 of the 64-byte row (chunk c of row r stored at c ^ {0,2,3,1}[(r >> 2) & 3]: conflict-free for this read
// pattern); TN: 16-lane group g transposes k-rows 8 g .. 8 g + 7 of 16 columns (chunk swizzle ((r & 3) << 2) | (((r >> 3) & 1) << 1)).
// Accumulators: 16x16 tiles of 4 registers, column l & 15, row 4 (l >> 4) + register.
template <class T, int EPI, bool STAMP = false, bool TN = false, bool M16 = false>
__global__ void __launch_bounds__(512) gemm_bf16_ring_kernel(const GemmBf16Params p, int ntiles, long long* stamps = nullptr) {
  static_assert(!TN || (T::BM == 256 && T::BN == 256), "the TN layout is built for the 256 x 256 tile");
  constexpr int MT16 = T::WTM / 16, NT16 = T::WTN / 16;
  constexpr int BM = T::BM, BN = T::BN, MT = T::MT, NTL = T::NTL, WTM = T::WTM, WTN = T::WTN;
  using R = Ring<T, EPI>;
  constexpr int NS = R::NS, D = R::D;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int wr = wave / T::WN, wc = wave % T::WN;
  // split-K (weight gradients: few output tiles, very long K): virtual tile v -> (k-slice v / tiles, output tile v % tiles), K-SLICE MAJOR:
  // the contiguous chunk of virtual tiles an XCD walks is then a few k-slices of EVERY output tile, so the rows of dY and X in those
  // slices are fetched once, by the one L2 that serves all their tiles.  (Until round 4 the order was tile major -- an XCD owned a few
  // tiles and all their slices, and every XCD streamed the whole of X: the fp32 twin of this kernel read 2.3x its operands that way,
  // profiles/r04_g_gemm_traffic_by_launch.txt.)  Slice z covers k in [z * kchunk, min(K, (z + 1) * kchunk)) and writes its partial
  // sums to slab z of C
  const int S = p.ksplit > 1 ? p.ksplit : 1;
  const int kchunk = S > 1 ? p.kchunk : ((p.K + 31) & ~31);
  // tile order inside an XCD's chunk: groups of GROUP_M row panels walked column by column, so that the ~32 tiles an XCD
  // runs at a time cover 8 row panels x 4 column panels (their A and B k-slices share the 4 MB L2) instead of 3-4 row
  // panels x every column panel (a 3.5-4.7 MB weight matrix alone overflows the L2)
  const int tiles_m = (p.M + BM - 1) / BM;
  auto tile_mn = [&](int t, int& m0, int& n0) {
    const int GROUP_M = p.group_m;
    const int per_group = GROUP_M * tiles_n, grp_i = t / per_group, within = t - grp_i * per_group;
    const int rows = tiles_m - grp_i * GROUP_M < GROUP_M ? tiles_m - grp_i * GROUP_M : GROUP_M;
    m0 = (grp_i * GROUP_M + within % rows) * BM;
    n0 = (within / rows) * BN;
  };
  auto slice_len = [&](int z) { const int rem = p.K - z * kchunk; return rem < kchunk ? rem : kchunk; };
  const int tiles_mn = tiles_m * tiles_n;
  const bool smaj = p.slice_major != 0;
  auto v_slice = [&](int vt) { return S > 1 ? (smaj ? vt / tiles_mn : vt % S) : 0; };
  auto v_tile = [&](int vt) { return S > 1 ? (smaj ? vt - (vt / tiles_mn) * tiles_mn : vt / S) : vt; };

  // ---- load side: runs D k-tiles ahead of the compute side, across tile boundaries ---------------------------
  // one DMA instruction fills 16 rows x 64 bytes: lane -> row lane >> 2, physical chunk lane & 3, which holds logical
  // chunk (lane & 3) ^ ((row >> 2) & 3); 16-row groups keep (row >> 2) & 3 == (lane >> 4) & 3
  // (TN: one instruction fills 2 k-rows x 512 bytes: lane -> row lane >> 5 of the pair, physical chunk lane & 31 holding
  //  logical chunk (lane & 31) ^ ((row & 3) << 2); wave w owns the row pairs w and w + 8, so row & 3 = 2 (w & 1) + (lane >> 5))
  const int srow = TN ? (lane >> 5) : (lane >> 2);
  constexpr unsigned STAB = 0x1320u;   // {0, 2, 3, 1}[i] = (STAB >> 4 i) & 3: chunk swizzle of the 16x16x32 row reads
  const int sc = TN ? ((lane & 31) ^ ((((2 * (wave & 1) + (lane >> 5)) & 3) << 2) | (M16 ? (((wave >> 2) & 1) << 1) : 0)))
                    : ((lane & 3) ^ (M16 ? (int)((STAB >> (4 * ((lane >> 4) & 3))) & 3u) : ((lane >> 4) & 3)));
  const unsigned offA = ((unsigned)(wave * (TN ? 2 : 16) + srow) * (unsigned)p.lda + sc * 8u) * 2u;
  const unsigned offB = ((unsigned)(wave * (TN ? 2 : 16) + srow) * (unsigned)p.ldb + sc * 8u) * 2u;
  // distance between a wave's DMA instructions of one k-tile: NT 128 tile rows, TN 16 k-rows
  const unsigned stepA = (TN ? 16u : 128u) * (unsigned)p.lda * 2u, stepB = (TN ? 16u : 128u) * (unsigned)p.ldb * 2u;
  int ltile = blockIdx.x, lt = 0, lslot = 0, lklen = 0, lnkt = 1;
  bool la_ok = true, lb_ok = true;   // TN: this lane's 8 columns lie inside the matrix (M, N % 8 == 0)
  __amdgpu_buffer_rsrc_t rsA, rsB;
  auto set_load_tile = [&](int v) {
    const int vt = xcd_chunk(v, ntiles), tile = v_tile(vt), k0 = v_slice(vt) * kchunk;
    int m0, n0;
    tile_mn(tile, m0, n0);
    lklen = slice_len(v_slice(vt));
    lnkt = (lklen + 31) / 32;
    long long abytes, bbytes;
    if (TN) {
      abytes = ((long long)(lklen - 1) * p.lda + (p.M - m0)) * 2;
      bbytes = ((long long)(lklen - 1) * p.ldb + (p.N - n0)) * 2;
      la_ok = m0 + sc * 8 < p.M;
      lb_ok = n0 + sc * 8 < p.N;
    } else {
      abytes = ((long long)(p.M - 1 - m0) * p.lda + lklen) * 2;
      bbytes = ((long long)(p.N - 1 - n0) * p.ldb + lklen) * 2;
    }
    if (abytes > 0x7FFFFFF0ll) abytes = 0x7FFFFFF0ll;
    if (bbytes > 0x7FFFFFF0ll) bbytes = 0x7FFFFFF0ll;
    const long long a0 = TN ? (long long)k0 * p.lda + m0 : (long long)m0 * p.lda + k0;
    const long long b0 = TN ? (long long)k0 * p.ldb + n0 : (long long)n0 * p.ldb + k0;
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A + a0), 0, (int)abytes, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B + b0), 0, (int)bbytes, 0x00020000);
  };
  set_load_tile(ltile);   // blockIdx.x < ntiles by construction of the grid
  auto issue_next = [&]() {
    // past the last tile the same instructions still issue (every lane out of range, zeros into a free slot): the
    // vmcnt bookkeeping then is the same for every stream length
    unsigned char* dst = smem + lslot * R::SLOT + wave * 1024;
    if constexpr (TN) {
      const bool live = ltile < ntiles;
      const unsigned base_a = offA + (unsigned)lt * 32u * (unsigned)p.lda * 2u, base_b = offB + (unsigned)lt * 32u * (unsigned)p.ldb * 2u;
#pragma unroll
      for (int i = 0; i < 2; ++i) {   // k-rows lt * 32 + 2 * wave + srow + 16 i
        const bool rok = live && lt * 32 + 2 * wave + srow + 16 * i < lklen;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + i * 8192), 16, (rok && la_ok) ? base_a + i * stepA : DGVIT_BUF_OOB, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool rok = live && lt * 32 + 2 * wave + srow + 16 * i < lklen;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(dst + 16384 + i * 8192), 16, (rok && lb_ok) ? base_b + i * stepB : DGVIT_BUF_OOB, 0, 0,
                                                 0);
      }
    } else {
      const unsigned dead = (ltile < ntiles && lt * 32 + sc * 8 < lklen) ? 0u : DGVIT_BUF_OOB;
      const unsigned oa = (offA + (unsigned)lt * 64u) | dead, ob = (offB + (unsigned)lt * 64u) | dead;
#pragma unroll
      for (int i = 0; i < R::GA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + i * 8192), 16, oa + i * stepA, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < R::GB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(dst + BM * 64 + i * 8192), 16, ob + i * stepB, 0, 0, 0);
    }
    lslot = lslot + 1 == NS ? 0 : lslot + 1;
    if (++lt == lnkt) {
      lt = 0;
      ltile += gridDim.x;
      if (ltile < ntiles) set_load_tile(ltile);
    }
  };

  // ---- compute side -------------------------------------------------------------------------------------------
  // fragment addresses inside a slot: row * 64 + ((2 s + h) ^ ((row >> 2) & 3)) * 16,  s = k-step 0 / 1
  const unsigned fsw = (unsigned)((li >> 2) & 3);
  const unsigned a_l0 = (unsigned)(wr * WTM + li) * 64u + ((h ^ fsw) * 16u);
  const unsigned b_l0 = (unsigned)(BM + wc * WTN + li) * 64u + ((h ^ fsw) * 16u);
  // TN transposed reads: lane 4q + p of a 16-lane group addresses row q of the 4 x 16 block, columns 4p .. 4p+3; the block of
  // k-step s, half u, 32-column tile X is rows 16 s + 8 h + 4 u + (0..3), columns 32 X + 16 ((lane >> 4) & 1) + (0..15)
  const unsigned tq = (unsigned)((lane & 15) >> 2), tp = (unsigned)(lane & 3), tcb = (unsigned)((lane >> 4) & 1);
  auto tr_base = [&](unsigned region, unsigned X) {   // byte address of (s = 0, u = 0) for 32-column tile X of an operand region
    return region + (8u * h + tq) * 512u + ((((X ^ tq) << 2) | (2u * tcb + (tp >> 1))) * 16u) + (tp & 1u) * 8u;
  };
  unsigned ta[TN ? MT : 1], tb[TN ? NTL : 1];
  if constexpr (TN) {
#pragma unroll
    for (int i = 0; i < MT; ++i) ta[i] = tr_base(0u, (unsigned)(wr * (WTM / 32) + i));
#pragma unroll
    for (int j = 0; j < NTL; ++j) tb[j] = tr_base(16384u, (unsigned)(wc * (WTN / 32) + j));
  }
  // M16 fragment bases
  const unsigned l15 = (unsigned)(lane & 15), g16 = (unsigned)(lane >> 4);
  const unsigned a16 = (unsigned)(wr * WTM + l15) * 64u + ((g16 ^ ((STAB >> (4 * (l15 >> 2))) & 3u)) * 16u);
  const unsigned b16 = (unsigned)(BM + wc * WTN + l15) * 64u + ((g16 ^ ((STAB >> (4 * (l15 >> 2))) & 3u)) * 16u);
  // TN + M16: group g16 reads k-rows 8 g16 + 4 u + (0..3) of the 16 columns of tile X16; lane 4q + p addresses row q, columns 4p..
  auto tr_base16 = [&](unsigned region, unsigned X16) {
    return region + (8u * g16 + tq) * 512u + ((((((X16 >> 1) ^ tq) << 2) | ((X16 & 1u) << 1) | (tp >> 1)) ^ ((g16 & 1u) << 1)) * 16u) +
           (tp & 1u) * 8u;
  };
  unsigned ta16[(TN && M16) ? MT16 : 1], tb16[(TN && M16) ? NT16 : 1];
  if constexpr (TN && M16) {
#pragma unroll
    for (int i = 0; i < MT16; ++i) ta16[i] = tr_base16(0u, (unsigned)(wr * MT16 + i));
#pragma unroll
    for (int j = 0; j < NT16; ++j) tb16[j] = tr_base16(16384u, (unsigned)(wc * NT16 + j));
  }
  auto tr_frag16 = [&](const unsigned char* sb, unsigned base) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sb + base));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sb + base + 2048));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto tr_frag = [&](const unsigned char* sb, unsigned base, int s) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sb + base + s * 8192));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sb + base + s * 8192 + 2048));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  float* es = reinterpret_cast<float*>(smem + NS * R::SLOT + wave * R::EPW);
  constexpr int LPR = WTN / 4, RPI = 64 / LPR;   // lanes per output row piece, rows per staging read
  const int erow = lane / LPR, ecol = (lane % LPR) * 4;

  f32x16 acc[M16 ? 1 : MT][M16 ? 1 : NTL];
  f32x4 acc4[M16 ? MT16 : 1][M16 ? NT16 : 1];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  int ctile = blockIdx.x, ct = 0, cslot = 0, since_epi = D, ntile_done = 0;
  int cnkt = (slice_len(v_slice(xcd_chunk(ctile, ntiles))) + 31) / 32;
  long long st0 = 0;
  if constexpr (STAMP) st0 = __builtin_amdgcn_s_memtime();

#pragma unroll
  for (int t = 0; t < D; ++t) issue_next();
  wait_vmcnt<R::WAIT>();
  __builtin_amdgcn_s_barrier();
  const int grp = wave >> 2;
  if (grp == 1) __builtin_amdgcn_s_barrier();
  while (ctile < ntiles) {
    const unsigned char* sb = smem + cslot * R::SLOT;
    bf16x8 af[M16 ? 1 : 2][M16 ? MT16 : MT], bf[M16 ? 1 : 2][M16 ? NT16 : NTL];
    if constexpr (M16) {
#pragma unroll
      for (int i = 0; i < MT16; ++i) {
        if constexpr (TN) af[0][i] = tr_frag16(sb, ta16[i]);
        else af[0][i] = *reinterpret_cast<const bf16x8*>(sb + a16 + i * 1024);
      }
#pragma unroll
      for (int j = 0; j < NT16; ++j) {
        if constexpr (TN) bf[0][j] = tr_frag16(sb, tb16[j]);
        else bf[0][j] = *reinterpret_cast<const bf16x8*>(sb + b16 + j * 1024);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if constexpr (TN) {
#pragma unroll
          for (int i = 0; i < MT; ++i) af[s][i] = tr_frag(sb, ta[i], s);
#pragma unroll
          for (int j = 0; j < NTL; ++j) bf[s][j] = tr_frag(sb, tb[j], s);
        } else {
#pragma unroll
          for (int i = 0; i < MT; ++i) af[s][i] = *reinterpret_cast<const bf16x8*>(sb + (a_l0 ^ (s * 32u)) + i * 2048);
#pragma unroll
          for (int j = 0; j < NTL; ++j) bf[s][j] = *reinterpret_cast<const bf16x8*>(sb + (b_l0 ^ (s * 32u)) + j * 2048);
        }
      }
    }
    issue_next();
    if (since_epi < D - 1) wait_vmcnt<R::WAIT_EPI>(); else wait_vmcnt<R::WAIT>();
    ++since_epi;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (M16) {
      if (ct == 0) {   // first k-tile of an output tile: accumulate onto zero
#pragma unroll
        for (int i = 0; i < MT16; ++i)
#pragma unroll
          for (int j = 0; j < NT16; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bf[0][j], zero4, 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < MT16; ++i)
#pragma unroll
          for (int j = 0; j < NT16; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bf[0][j], acc4[i][j], 0, 0, 0);
      }
    } else {
      if (ct == 0) {   // first k-tile of an output tile: accumulate onto zero (no 128-register clear)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], zero16, 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    cslot = cslot + 1 == NS ? 0 : cslot + 1;
    if (++ct < cnkt) continue;

    // ---- epilogue of tile `ctile`: acc (column on the lane, rows in registers) -> wave-private LDS -> row pieces ----
    long long st1 = 0;
    if constexpr (STAMP) st1 = __builtin_amdgcn_s_memtime();
    {
      const int vt = xcd_chunk(ctile, ntiles), tile = v_tile(vt);
      int m0, n0;
      tile_mn(tile, m0, n0);
      const int gn = n0 + wc * WTN + ecol;
      const bool ncol = gn < p.N;
      // tile-relative buffer addressing: offsets stay small, invalid rows / columns are sent out of range
      const long long crow0 = p.c_rgrp > 0 ? m0 + m0 / p.c_rgrp + 1 : m0;
      constexpr int CES = (EPI == BEPI_F32 || EPI == BEPI_F32_PLAIN) ? 4 : 2;
      const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<unsigned char*>(p.C) + ((long long)v_slice(vt) * p.slab_stride + crow0 * p.ldc + n0) * CES, 0, (int)DGVIT_BUF_OOB,
          0x00020000);
      __amdgpu_buffer_rsrc_t rsX = rsC;   // second operand of the epilogue: residual (fp32) / pre-activation copy / aux (bf16)
      if (EPI == BEPI_F32 && p.res)
        rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res) + ((long long)(p.res_mod > 0 ? 0 : m0) * p.ldr + n0), 0,
                                                (int)DGVIT_BUF_OOB, 0x00020000);
      if (EPI == BEPI_GELU2_BF16)
        rsX = __builtin_amdgcn_make_buffer_rsrc(p.C2 + ((long long)m0 * p.ldc2 + n0), 0, (int)DGVIT_BUF_OOB, 0x00020000);
      if (EPI == BEPI_DGELU_BF16)
        rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.aux) + ((long long)m0 * p.ldaux + n0), 0, (int)DGVIT_BUF_OOB,
                                                0x00020000);
      fx4 bias4 = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && ncol) bias4 = *reinterpret_cast<const fx4*>(p.bias + gn);
      const unsigned coln = (unsigned)(wc * WTN + ecol);
      // GELU' epilogue: the pre-activation loads of 32-row group i + 1 are issued before group i is processed (the
      // fragment registers are free now), so their latency -- they queue behind the LDS-DMAs already in flight -- is
      // paid about once per tile, not once per store
      constexpr int NQ = 16 / RPI;
      u32x2v auxv[EPI == BEPI_DGELU_BF16 ? MT : 1][2 * NQ];
      auto aux_fetch = [&](int i, u32x2v (&dst)[2 * NQ]) {
#pragma unroll
        for (int hq = 0; hq < 2 * NQ; ++hq) {
          const int ml = wr * WTM + i * 32 + (hq / NQ) * 16 + (hq % NQ) * RPI + erow;
          const unsigned xoff = (m0 + ml < p.M && ncol) ? ((unsigned)ml * (unsigned)p.ldaux + coln) * 2u : DGVIT_BUF_OOB;
          dst[hq] = __builtin_amdgcn_raw_buffer_load_b64(rsX, xoff, 0, 0);
        }
      };
      if (EPI == BEPI_DGELU_BF16) aux_fetch(0, auxv[0]);
      // residual epilogue: the same look-ahead.  A load issued inside the piece loop makes every piece wait vmcnt(0) - the load
      // AND the previous piece's store, one loaded-memory round trip per 16-byte piece, 32 pieces per thread and tile: the
      // out-projection (K = 768: ~10 us of MFMA work per tile) spent most of its time there.  The residual of 16-row group g + 1 is
      // requested piece by piece as group g consumes its own (4 NQ registers): a load is always older than the stores behind it.
      const bool use_res = EPI == BEPI_F32 && p.res != nullptr;
      fx4 resv[EPI == BEPI_F32 ? NQ : 1];
      auto res_load = [&](int g, int q) -> fx4 {   // g = 2 i + half: one 16-row group of the wave's tile; q: its q-th row piece
        const int ml = wr * WTM + g * 16 + q * RPI + erow, gm = m0 + ml;
        const unsigned rr = p.res_mod > 0 ? (unsigned)(gm % p.res_mod) + 1u : (unsigned)ml;
        const unsigned roff = (gm < p.M && ncol) ? (rr * (unsigned)p.ldr + coln) * 4u : DGVIT_BUF_OOB;
        return __builtin_bit_cast(fx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, roff, 0, 0));
      };
      if (use_res) {
#pragma unroll
        for (int q = 0; q < (EPI == BEPI_F32 ? NQ : 1); ++q) resv[q] = res_load(0, q);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (EPI == BEPI_DGELU_BF16 && half == 0 && i + 1 < MT) aux_fetch(i + 1, auxv[EPI == BEPI_DGELU_BF16 ? i + 1 : 0]);
          __builtin_amdgcn_wave_barrier();
          if constexpr (M16) {   // 16-row block 2 i + half: rows 4 (lane >> 4) + r, columns 16 j + (lane & 15)
#pragma unroll
            for (int j = 0; j < NT16; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) es[(4 * (lane >> 4) + r) * WTN + j * 16 + (lane & 15)] = acc4[2 * i + half][j][r];
          } else {
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
              for (int r = 0; r < 8; ++r) es[((r & 3) + 8 * (r >> 2) + 4 * h) * WTN + j * 32 + li] = acc[i][j][8 * half + r];
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int q = 0; q < 16 / RPI; ++q) {
            const int lr = q * RPI + erow;
            fx4 v = *reinterpret_cast<const fx4*>(es + lr * WTN + ecol);
            const int ml = wr * WTM + i * 32 + half * 16 + lr, gm = m0 + ml;   // row inside the tile / global
            const bool ok = gm < p.M && ncol;
            v += bias4;
            const long long crow = p.c_rgrp > 0 ? gm + gm / p.c_rgrp + 1 : gm;
            const unsigned coff = ok ? ((unsigned)(crow - crow0) * (unsigned)p.ldc + coln) * CES : DGVIT_BUF_OOB;
            if (EPI == BEPI_F32 || EPI == BEPI_F32_PLAIN) {
              if (use_res) {   // this piece's residual arrived a group ago; its register takes the next group's piece at once
                v += resv[EPI == BEPI_F32 ? q : 0];
                if (2 * i + half + 1 < 2 * MT) resv[EPI == BEPI_F32 ? q : 0] = res_load(2 * i + half + 1, q);
              }
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), rsC, coff, 0, 0);
            } else if (EPI == BEPI_BF16) {
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, __builtin_convertvector(v, bf16x4)), rsC, coff, 0, 0);
            } else if (EPI == BEPI_GELU_BF16 || EPI == BEPI_GELU2_BF16) {
              if (EPI == BEPI_GELU2_BF16)   // ldc2 == ldc (checked at launch): the pre-activation copy shares the tile-relative offset
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, __builtin_convertvector(v, bf16x4)), rsX, coff, 0, 0);
              fx4 g = gelu_bf16x4(v);
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, __builtin_convertvector(g, bf16x4)), rsC, coff, 0, 0);
            } else {  // BEPI_DGELU_BF16
              const bf16x4 a = __builtin_bit_cast(bf16x4, auxv[EPI == BEPI_DGELU_BF16 ? i : 0][half * NQ + q]);
              fx4 g = {v[0] * gelu_erf_grad((float)a[0]), v[1] * gelu_erf_grad((float)a[1]), v[2] * gelu_erf_grad((float)a[2]),
                       v[3] * gelu_erf_grad((float)a[3])};
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, __builtin_convertvector(g, bf16x4)), rsC, coff, 0, 0);
            }
          }
        }
    }
    if constexpr (STAMP) {
      if (lane == 0 && (wave & 3) == 0 && ntile_done < 8) {
        long long* o = stamps + (((long long)blockIdx.x * 2 + grp) * 8 + ntile_done) * 4;
        o[0] = st0; o[1] = st1; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
      }
      st0 = __builtin_amdgcn_s_memtime();
    }
    ++ntile_done;
    ct = 0;
    since_epi = 0;
    ctile += gridDim.x;
    if (ctile < ntiles) cnkt = (slice_len(v_slice(xcd_chunk(ctile, ntiles))) + 31) / 32;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();   // the trailing (all-zero) LDS-DMAs must land before the workgroup gives its LDS back
}

int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

// one ring-kernel instantiation: raise its dynamic-LDS limit on first use on this device, then launch it
template <class T, int EPI, bool STAMP, bool TN, bool M16>
int launch_ring(const GemmBf16Params& p, int grid, int vtiles, long long* stamps, hipStream_t st) {
  constexpr int LDS = Ring<T, EPI>::LDS;
  auto kern = gemm_bf16_ring_kernel<T, EPI, STAMP, TN, M16>;
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "gemm_bf16: cannot raise the dynamic LDS limit to %d bytes", LDS);
    once.mark(bit);
  }
  const int slot = STAMP ? -1 : profile_begin(PROF_GEMM, 2.0 * p.M * p.N * p.K, st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, p, vtiles, stamps);
  if (!STAMP) profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("gemm_bf16_ring_kernel");
  return DGVIT_OK;
}

template <class T, int EPI, bool RING>
int launch(const GemmBf16Params& p_in, hipStream_t st) {
  GemmBf16Params p = p_in;
  if (p.group_m <= 0) p.group_m = g_gemm_bf16_group_m > 0 ? g_gemm_bf16_group_m : 8;
  p.slice_major = g_gemm_zfold;
  const long long tiles = (long long)((p.M + T::BM - 1) / T::BM) * ((p.N + T::BN - 1) / T::BN);
  DGVIT_CHECK_ARG(tiles < (1ll << 30), "gemm_bf16: too many tiles");
  if constexpr (RING) {
    // tile-relative 32-bit buffer offsets in the epilogue
    DGVIT_CHECK_ARG((long long)(T::BM + T::BM / (p.c_rgrp > 0 ? p.c_rgrp : T::BM) + 2) * p.ldc * 4 < (1ll << 31) &&
                        (long long)(p.res_mod > 0 ? p.res_mod + 1 : T::BM) * p.ldr * 4 < (1ll << 31) &&
                        (long long)T::BM * p.ldc2 * 2 < (1ll << 31) && (long long)T::BM * p.ldaux * 2 < (1ll << 31),
                    "gemm_bf16: output leading dimension too large");
    DGVIT_CHECK_ARG(EPI != BEPI_GELU2_BF16 || (p.ldc2 == p.ldc && p.c_rgrp == 0), "gemm_bf16: the GELU epilogue with a pre-activation copy needs ldc2 == ldc");
    const int S = p.ksplit > 1 ? p.ksplit : 1;
    DGVIT_CHECK_ARG(S == 1 || (p.kchunk > 0 && p.kchunk % 32 == 0 && (long long)(S - 1) * p.kchunk < p.K && (long long)S * p.kchunk >= p.K),
                    "gemm_bf16: bad split-K plan (%d x %d over K=%d)", S, p.kchunk, p.K);
    DGVIT_CHECK_ARG(S == 1 || EPI == BEPI_F32_PLAIN, "gemm_bf16: split-K needs the plain fp32 epilogue");
    const long long vtiles = tiles * S;
    DGVIT_CHECK_ARG(vtiles < (1ll << 30), "gemm_bf16: too many tiles");
    const int grid = (int)(vtiles < num_cus() ? vtiles : num_cus());   // one persistent workgroup per CU
#ifdef DGVIT_DIAG
    if constexpr (EPI == BEPI_BF16) {
      if (g_gemm_bf16_stamps) return launch_ring<T, EPI, true, false, false>(p, grid, (int)vtiles, g_gemm_bf16_stamps, st);
    }
#endif
    if constexpr (EPI == BEPI_F32_PLAIN && T::BM == 256 && T::BN == 256) {
      if (p.tn) {   // contraction-major operands (weight gradients)
        KNOB_IF(g_gemm_bf16_m16) return launch_ring<T, EPI, false, true, true>(p, grid, (int)vtiles, nullptr, st);
        else return launch_ring<T, EPI, false, true, false>(p, grid, (int)vtiles, nullptr, st);
      }
    }
    DGVIT_CHECK_ARG(!p.tn, "gemm_bf16: the TN layout needs the plain fp32 epilogue and the 256 x 256 tile");
    KNOB_IF(g_gemm_bf16_m16) return launch_ring<T, EPI, false, false, true>(p, grid, (int)vtiles, nullptr, st);
    else return launch_ring<T, EPI, false, false, false>(p, grid, (int)vtiles, nullptr, st);
  } else {
    DGVIT_CHECK_ARG(p.ksplit <= 1 && !p.tn, "gemm_bf16: split-K / TN need the 256 x 256 tile");
    static DeviceOnce once;
    if (const unsigned long long bit = once.pending()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<T, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              T::LDS) != hipSuccess)
        return dgvit_set_error(DGVIT_ERR_HIP, "gemm_bf16: cannot raise the dynamic LDS limit to %d bytes", T::LDS);
      once.mark(bit);
    }
    const int slot = profile_begin(PROF_GEMM, 2.0 * p.M * p.N * p.K, st);
    hipLaunchKernelGGL((gemm_bf16_kernel<T, EPI>), dim3((unsigned)tiles), dim3(T::NT), T::LDS, st, p);
    profile_end(slot, st);
  }
  DGVIT_CHECK_LAUNCH("gemm_bf16_kernel");
  return DGVIT_OK;
}

template <int EPI>
int dispatch(const GemmBf16Params& p, hipStream_t st) {
  // tile choice: 256x256 ring kernel when the grid still fills the chip, else the 128x128 simple kernel
  int tile = g_gemm_bf16_tile_hint;
  if (!tile) {
    const long long t256 = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    tile = (p.ksplit > 1 || p.tn || (p.M >= 256 && p.N >= 256 && t256 >= 512)) ? 256256 : 128128;
    // a handful of 128 x 128 tiles with a long K (the token-0-only last block at inference: 440 x 768 x 3072 = 24 tiles) leaves nine CUs
    // in ten idle behind a serial k-loop: 64 x 64 tiles give four times the workgroups
    if (tile == 128128 && (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) < 128) tile = 64064;
    if (tile == 256256 && gemm_bf16_stream_supports(EPI, p)) tile = 256257;   // the stream kernel (gemm_bf16_stream.hip) where it applies
  }
  switch (tile) {
    case 256256: return launch<BTile<256, 256, 2, 4>, EPI, true>(p, st);
    case 256128: return launch<BTile<256, 128, 4, 2>, EPI, true>(p, st);
    case 128128: return launch<BTile<128, 128, 2, 2>, EPI, false>(p, st);
    case 64064: return launch<BTile<64, 64, 2, 2>, EPI, false>(p, st);
    case 256257:   // (a hint for a problem the stream kernel does not take falls back to the ring kernel)
      if (!gemm_bf16_stream_supports(EPI, p)) return launch<BTile<256, 256, 2, 4>, EPI, true>(p, st);
      return gemm_bf16_stream(EPI, p, st);
#ifdef DGVIT_DIAG
    case 256254: return launch<BTile<256, 256, 2, 2>, EPI, false>(p, st);   // experiment: 4 waves of 128 x 128 (one per SIMD), per-tile kernel
#endif
    default: return dgvit_set_error(DGVIT_ERR_ARG, "gemm_bf16: unknown tile %d", tile);
  }
}

}  // namespace

int gemm_bf16(int epi, const GemmBf16Params& p, hipStream_t st) {
  DGVIT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm_bf16: empty problem %d x %d x %d", p.M, p.N, p.K);
  DGVIT_CHECK_ARG(p.lda % 8 == 0 && p.ldb % 8 == 0 && (p.tn ? (p.M % 8 == 0 && p.N % 8 == 0) : p.K % 8 == 0),
                  "gemm_bf16: leading dimensions and the contiguous extents (K; TN: M, N) must be multiples of 8 (16-byte chunks)");
  DGVIT_CHECK_ARG(p.N % 4 == 0 && p.ldc % 4 == 0, "gemm_bf16: N and ldc must be multiples of 4");
  DGVIT_CHECK_ARG(((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C | (uintptr_t)p.res | (uintptr_t)p.bias | (uintptr_t)p.C2 | (uintptr_t)p.aux) % 16 == 0,
                  "gemm_bf16: operands must be 16-byte aligned");
  DGVIT_CHECK_ARG((!p.res || p.ldr % 4 == 0) && (epi != BEPI_DGELU_BF16 || (p.aux && p.ldaux % 4 == 0)), "gemm_bf16: bad residual / aux");
  DGVIT_CHECK_ARG((long long)p.lda * 2 * 256 < (1ll << 30) && (long long)p.ldb * 2 * 256 < (1ll << 30), "gemm_bf16: leading dimension too large");
  DGVIT_CHECK_ARG(!p.tn || (long long)(p.ksplit > 1 ? p.kchunk : p.K) * (p.lda > p.ldb ? p.lda : p.ldb) * 2 < (1ll << 31),
                  "gemm_bf16: TN k-slice too long for 32-bit buffer offsets (split K further)");
  switch (epi) {
    case BEPI_BF16: return dispatch<BEPI_BF16>(p, st);
    case BEPI_GELU_BF16: return dispatch<BEPI_GELU_BF16>(p, st);
    case BEPI_F32: return dispatch<BEPI_F32>(p, st);
    case BEPI_DGELU_BF16: return dispatch<BEPI_DGELU_BF16>(p, st);
    case BEPI_F32_PLAIN: return dispatch<BEPI_F32_PLAIN>(p, st);
    case BEPI_GELU2_BF16:
      DGVIT_CHECK_ARG(p.C2 && p.ldc2 % 4 == 0, "gemm_bf16: epilogue 5 needs C2");
      return dispatch<BEPI_GELU2_BF16>(p, st);
    default: return dgvit_set_error(DGVIT_ERR_ARG, "gemm_bf16: unknown epilogue %d", epi);
  }
}
