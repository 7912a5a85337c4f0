// bf16 GEMM for gfx950:  C[m][n] = sum_k A[m][k] * B[n][k]   (A, B bf16 with k contiguous; fp32 accumulate on
// v_mfma_f32_32x32x16_bf16; fused bias / residual / erf-GELU epilogues).  The Linear layers of Attention and
// FeedForward (GoalFormer.py:42-50,64,66-69) in the bf16 configuration (BASELINE config 5).
//
// Structure (one workgroup per BM x BN output tile, BK = 64):
//  * Staging is LDS-DMA: `buffer_load_dwordx4 ... lds` writes 16 bytes per lane straight into LDS (no VGPRs, no
//    ds_write).  One wave instruction fills 8 tile rows x 128 bytes.  The LDS image is row-major with 128-byte rows
//    whose eight 16-byte chunks are XOR-swizzled by ((row >> 1) & 7): an LDS-DMA writes lane-linearly, so the swizzle
//    is applied to the per-lane SOURCE address, and again by the fragment reads.  With it every ds_read_b128 lane
//    group {0-3,12-15,20-27} / {4-11,16-19,28-31} touches 16 distinct 16-byte bank slots (conflict-free).
//  * Rows beyond M / N and the k tail are zero-filled by the buffer range check (offset >= num_records reads 0),
//    so the main loop is branch-free.
//  * Two LDS buffers: the DMA of k-tile t+1 is issued before the MFMAs of tile t and waited for after them
//    (one barrier per k-tile).
//  * Each wave owns a (BM/WM) x (BN/WN) block of 32x32 accumulators; an A/B fragment is one ds_read_b128
//    (lane (i, h): row i, k = 16 s + 8 h .. + 7).
//  * Epilogue: accumulators (column on the lane, rows in registers) are transposed through a wave-private slice of
//    the now idle staging LDS so that global accesses are 16-byte (fp32) / 8-byte (bf16) row-contiguous pieces.
#include "bf16.h"
#include "kernels.h"

int g_gemm_bf16_tile_hint = 0;

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

  // ---- LDS-DMA plan ------------------------------------------------------------------------------------
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

  // ---- fragment addresses (bytes inside a buffer): row * 128 + ((2 s + h) ^ f(row)) * 16 ------------------
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

  // ---- epilogue -------------------------------------------------------------------------------------------
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
        } else if (EPI == BEPI_GELU_BF16) {
          if (p.C2) *reinterpret_cast<bf16x4*>(p.C2 + (long long)gm * p.ldc2 + gn) = __builtin_convertvector(v, bf16x4);
          fx4 g = {gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
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

template <class T, int EPI>
int launch(const GemmBf16Params& p, hipStream_t st) {
  static bool attr_done = false;   // the kernels use more than the 64 KB default dynamic LDS limit
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<T, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            T::LDS) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "gemm_bf16: cannot raise the dynamic LDS limit to %d bytes", T::LDS);
    attr_done = true;
  }
  const long long tiles = (long long)((p.M + T::BM - 1) / T::BM) * ((p.N + T::BN - 1) / T::BN);
  const int slot = profile_begin(PROF_GEMM, 2.0 * p.M * p.N * p.K, st);
  hipLaunchKernelGGL((gemm_bf16_kernel<T, EPI>), dim3((unsigned)tiles), dim3(T::NT), T::LDS, st, p);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("gemm_bf16_kernel");
  return DGVIT_OK;
}

template <int EPI>
int dispatch(const GemmBf16Params& p, hipStream_t st) {
  // tile choice: 256x256 (8 waves) when the grid still fills the chip several times over, else 128x128 (4 waves)
  int tile = g_gemm_bf16_tile_hint;
  if (!tile) {
    const long long t256 = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    tile = (p.M >= 256 && p.N >= 256 && t256 >= 512) ? 256256 : 128128;
  }
  switch (tile) {
    case 256256: return launch<BTile<256, 256, 2, 4>, EPI>(p, st);
    case 256128: return launch<BTile<256, 128, 4, 2>, EPI>(p, st);
    case 128128: return launch<BTile<128, 128, 2, 2>, EPI>(p, st);
    default: return dgvit_set_error(DGVIT_ERR_ARG, "gemm_bf16: unknown tile %d", tile);
  }
}

}  // namespace

int gemm_bf16(int epi, const GemmBf16Params& p, hipStream_t st) {
  DGVIT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm_bf16: empty problem %d x %d x %d", p.M, p.N, p.K);
  DGVIT_CHECK_ARG(p.K % 8 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0, "gemm_bf16: K, lda, ldb must be multiples of 8 (16-byte rows)");
  DGVIT_CHECK_ARG(p.N % 4 == 0 && p.ldc % 4 == 0, "gemm_bf16: N and ldc must be multiples of 4");
  DGVIT_CHECK_ARG(((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) % 16 == 0, "gemm_bf16: operands must be 16-byte aligned");
  DGVIT_CHECK_ARG((long long)p.lda * 2 * 256 < (1ll << 30) && (long long)p.ldb * 2 * 256 < (1ll << 30), "gemm_bf16: leading dimension too large");
  switch (epi) {
    case BEPI_BF16: return dispatch<BEPI_BF16>(p, st);
    case BEPI_GELU_BF16: return dispatch<BEPI_GELU_BF16>(p, st);
    case BEPI_F32: return dispatch<BEPI_F32>(p, st);
    case BEPI_DGELU_BF16: return dispatch<BEPI_DGELU_BF16>(p, st);
    case BEPI_F32_PLAIN: return dispatch<BEPI_F32_PLAIN>(p, st);
    default: return dgvit_set_error(DGVIT_ERR_ARG, "gemm_bf16: unknown epilogue %d", epi);
  }
}
