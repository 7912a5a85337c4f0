// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 in, fp32 accumulate).
//
// This is the dominant kernel of the DGViT hot path: every Linear of the encoder
// (GoalFormer.py:43,46,64,67,139 and the head Linears of got_sac_network.py) and both of its
// gradients run through it.  One kernel template covers the three operand layouts
//   NT  Y  = X W^T      (forward)         A k-contiguous, B k-contiguous
//   NN  dX = dY W       (data gradient)   A k-contiguous, B n-contiguous
//   TN  dW = dY^T X     (weight gradient) A m-contiguous, B n-contiguous, split over K = tokens
// without any transposed copy in HBM:
//   * a k-contiguous operand is staged as LDS[row][BK+4] and read back with one ds_read_b128 per
//     32-row MFMA tile and 8-deep k-group (lane (i, h) takes k = 8g + 4h .. +3; the +4 pad makes the
//     16-lane b128 groups hit 16 distinct 16-byte bank slots);
//   * an m/n-contiguous operand is staged as LDS[k][R+4] and read with ds_read_b32 (32 consecutive
//     floats per half-wave, conflict free);
//   both read paths feed MFMA step s of k-group g with k = 8g + 4h + s, so the contraction order is the
//   same permutation for A and B whatever their layouts.
// Global->LDS staging is register-staged and double-buffered in LDS: tile t+1 is fetched to VGPRs
// before the MFMAs of tile t issue and written to the other LDS buffer after them (one barrier per
// k-tile).  Four waves (2x2) per workgroup, each owning a (BM/2)x(BN/2) block of 32x32 accumulators.
#include "common.h"

namespace {

template <int BM_, int BN_, int BK_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_;
};

__device__ __forceinline__ int xcd_remap(int id, int n) {
  // Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous chunk of the tile grid so
  // that the column tiles sharing an A row-panel hit the same L2 (speed only, bijective for any n).
  const int q = n >> 3, r = n & 7, xcd = id & 7, loc = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// ---- global -> register tile fetch ---------------------------------------------------------------
// KC: tile is R rows x BK k (k contiguous in memory).  MC: tile is BK k-rows x R (row index contiguous).
template <int R, int BK, bool KC, int VEC>
struct Fetch {
  static constexpr int NV = R * BK / 4 / 256;  // float4 slots per thread
  static constexpr int PER_ROW = KC ? BK / 4 : R / 4;

  __device__ static __forceinline__ void run(float4 (&reg)[NV], const float* __restrict__ base, int ld, int r0,
                                             int rmax, int k0, int kend, int tid, int kgrp) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * 256;
      const int a = f / PER_ROW, c = (f % PER_ROW) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KC) {
        const int row = r0 + a, k = k0 + c;
        if (row < rmax) {
          const float* src = base + (long long)row * ld + k;
          if (VEC == 4) {
            if (k < kend) v = *reinterpret_cast<const float4*>(src);
          } else {
            if (k + 0 < kend) v.x = src[0];
            if (k + 1 < kend) v.y = src[1];
            if (k + 2 < kend) v.z = src[2];
            if (k + 3 < kend) v.w = src[3];
          }
        }
      } else {
        const int k = k0 + a, col = r0 + c;
        if (k < kend) {
          const long long prow = kgrp > 0 ? (long long)k + k / kgrp + 1 : (long long)k;
          const float* src = base + prow * ld + col;
          if (VEC == 4) {
            if (col < rmax) v = *reinterpret_cast<const float4*>(src);
          } else {
            if (col + 0 < rmax) v.x = src[0];
            if (col + 1 < rmax) v.y = src[1];
            if (col + 2 < rmax) v.z = src[2];
            if (col + 3 < rmax) v.w = src[3];
          }
        }
      }
      reg[i] = v;
    }
  }

  __device__ static __forceinline__ void stash(const float4 (&reg)[NV], float* lds, int tid) {
    constexpr int STRIDE = KC ? BK + 4 : R + 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * 256;
      const int a = f / PER_ROW, c = (f % PER_ROW) * 4;
      *reinterpret_cast<float4*>(lds + a * STRIDE + c) = reg[i];
    }
  }
};

// ---- LDS -> MFMA operand fragments for one 8-deep k-group -----------------------------------------
template <int R, int BK, bool KC>
__device__ __forceinline__ void frag(float (&out)[4], const float* lds, int row, int g, int h) {
  if (KC) {
    const float4 v = *reinterpret_cast<const float4*>(lds + row * (BK + 4) + 8 * g + 4 * h);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else {
    const float* p = lds + (8 * g + 4 * h) * (R + 4) + row;
#pragma unroll
    for (int s = 0; s < 4; ++s) out[s] = p[s * (R + 4)];
  }
}

template <class T, int LAYOUT, int VEC, int EPI>
__global__ void __launch_bounds__(256, 2) gemm_f32_kernel(const GemmParams p) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK;
  constexpr bool AKC = LAYOUT != GEMM_TN;
  constexpr bool BKC = LAYOUT == GEMM_NT;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int A_TILE = AKC ? BM * (BK + 4) : BK * (BM + 4);
  constexpr int B_TILE = BKC ? BN * (BK + 4) : BK * (BN + 4);
  constexpr int STAGE = A_TILE + B_TILE;
  using FA = Fetch<BM, BK, AKC, VEC>;
  using FB = Fetch<BN, BK, BKC, VEC>;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kbeg = blockIdx.z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[FA::NV], rb[FB::NV];
  FA::run(ra, p.A, p.lda, m0, p.M, kbeg, kend, tid, p.a_kgrp);
  FB::run(rb, p.B, p.ldb, n0, p.N, kbeg, kend, tid, 0);
  FA::stash(ra, smem, tid);
  FB::stash(rb, smem + A_TILE, tid);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const float* la = smem + (kt & 1) * STAGE;
    const float* lb = la + A_TILE;
    const bool more = kt + 1 < nk;
    if (more) {
      FA::run(ra, p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, tid, p.a_kgrp);
      FB::run(rb, p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, tid, 0);
    }
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float fa[TM][4], fb[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) frag<BM, BK, AKC>(fa[i], la, wm * WM + i * 32 + li, g, h);
#pragma unroll
      for (int j = 0; j < TN; ++j) frag<BN, BK, BKC>(fb[j], lb, wn * WN + j * 32 + li, g, h);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    }
    if (more) {
      float* wa = smem + ((kt + 1) & 1) * STAGE;
      FA::stash(ra, wa, tid);
      FB::stash(rb, wa + A_TILE, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: accumulator (col = lane&31, row = (r&3) + 8*(r>>2) + 4*h) -> global ---------------
  float* Cz = p.C;
  if (EPI == EPI_SPLITK) Cz += (long long)blockIdx.z * p.slab_stride;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 32 + li;
    if (n >= p.N) continue;
    float bias = 0.f;
    if ((EPI == EPI_STORE || EPI == EPI_GELU2 || EPI == EPI_RELU) && p.bias) bias = p.bias[n];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        if (EPI == EPI_STORE) {
          v += bias;
          if (p.res) {
            const long long rr = p.res_mod > 0 ? (long long)(m % p.res_mod) + 1 : (long long)m;
            v += p.res[rr * p.ldr + n];
          }
          const long long cm = p.c_rgrp > 0 ? (long long)m + m / p.c_rgrp + 1 : (long long)m;
          Cz[cm * p.ldc + n] = v;
        } else if (EPI == EPI_GELU2) {
          v += bias;
          Cz[(long long)m * p.ldc + n] = v;
          p.C2[(long long)m * p.ldc2 + n] = gelu_erf(v);
        } else if (EPI == EPI_DGELU) {
          Cz[(long long)m * p.ldc + n] = v * gelu_erf_grad(p.aux[(long long)m * p.ldaux + n]);
        } else if (EPI == EPI_RELU) {
          Cz[(long long)m * p.ldc + n] = fmaxf(v + bias, 0.f);
        } else if (EPI == EPI_DRELU) {
          Cz[(long long)m * p.ldc + n] = p.aux[(long long)m * p.ldaux + n] > 0.f ? v : 0.f;
        } else {
          Cz[(long long)m * p.ldc + n] = v;
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256) reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                           long long n4, int nslab, long long stride) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 s = reinterpret_cast<const float4*>(slabs)[i];
  for (int z = 1; z < nslab; ++z) {
    const float4 v = reinterpret_cast<const float4*>(slabs + z * stride)[i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(out)[i] = s;
}

__global__ void __launch_bounds__(256) reduce_slabs_scalar_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                                  long long n, int nslab, long long stride) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = slabs[i];
  for (int z = 1; z < nslab; ++z) s += slabs[z * stride + i];
  out[i] = s;
}

template <class T, int LAYOUT, int VEC, int EPI>
int launch(const GemmParams& p, int nsplit, hipStream_t stream) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK;
  constexpr bool AKC = LAYOUT != GEMM_TN;
  constexpr bool BKC = LAYOUT == GEMM_NT;
  constexpr int A_TILE = AKC ? BM * (BK + 4) : BK * (BM + 4);
  constexpr int B_TILE = BKC ? BN * (BK + 4) : BK * (BN + 4);
  constexpr size_t lds = 2 * (A_TILE + B_TILE) * sizeof(float);
  static bool attr_done = false;
  auto kern = gemm_f32_kernel<T, LAYOUT, VEC, EPI>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  const long long tiles = (long long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  DGVIT_CHECK_ARG(tiles > 0 && tiles < (1ll << 31), "gemm: bad tile count %lld", tiles);
  dim3 grid((unsigned)tiles, 1, (unsigned)nsplit);
  const int slot = profile_begin(PROF_GEMM, 2.0 * p.M * p.N * p.K, stream);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("gemm_f32_kernel");
  return DGVIT_OK;
}

using T128 = TileCfg<128, 128, 32>;
using T64 = TileCfg<64, 64, 32>;

template <int LAYOUT, int EPI>
int pick_tile(const GemmParams& p, int nsplit, bool vec4, int tile_hint, hipStream_t stream) {
  if (!vec4) return launch<T64, LAYOUT, 1, EPI>(p, nsplit, stream);
  bool big = p.M >= 128 && p.N >= 128;
  if (tile_hint == 64) big = false;
  if (tile_hint == 128) big = true;
  return big ? launch<T128, LAYOUT, 4, EPI>(p, nsplit, stream) : launch<T64, LAYOUT, 4, EPI>(p, nsplit, stream);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

int g_gemm_tile_hint = 0;  // test/bench override: 0 auto, 64, 128

int gemm_f32(int layout, int epi, const GemmParams& p, int nsplit, hipStream_t stream) {
  DGVIT_CHECK_ARG(p.A && p.B && p.C, "gemm: null operand");
  DGVIT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  DGVIT_CHECK_ARG(nsplit >= 1 && p.kchunk > 0 && p.kchunk % 32 == 0, "gemm: kchunk must be a positive multiple of 32");
  DGVIT_CHECK_ARG((long long)p.kchunk * nsplit >= p.K, "gemm: split does not cover K");
  // float4 staging needs 16-byte aligned bases, leading dims that are multiples of 4 and a
  // contiguous extent that is a multiple of 4; anything else takes the scalar-load 64x64 variant.
  bool vec4 = al16(p.A) && al16(p.B) && p.lda % 4 == 0 && p.ldb % 4 == 0;
  if (layout == GEMM_NT) vec4 = vec4 && p.K % 4 == 0;
  if (layout == GEMM_NN) vec4 = vec4 && p.K % 4 == 0 && p.N % 4 == 0;
  if (layout == GEMM_TN) vec4 = vec4 && p.M % 4 == 0 && p.N % 4 == 0;
  const int th = g_gemm_tile_hint;
#define CASE(L, E) \
  if (layout == L && epi == E) return pick_tile<L, E>(p, nsplit, vec4, th, stream);
  CASE(GEMM_NT, EPI_STORE)
  CASE(GEMM_NT, EPI_GELU2)
  CASE(GEMM_NT, EPI_RELU)
  CASE(GEMM_NN, EPI_STORE)
  CASE(GEMM_NN, EPI_DGELU)
  CASE(GEMM_NN, EPI_DRELU)
  CASE(GEMM_TN, EPI_SPLITK)
#undef CASE
  return dgvit_set_error(DGVIT_ERR_ARG, "gemm: unsupported layout/epilogue %d/%d", layout, epi);
}

int reduce_slabs(const float* slabs, float* out, long long n, int nslab, long long slab_stride, hipStream_t stream) {
  DGVIT_CHECK_ARG(slabs && out && n > 0 && nslab >= 1, "reduce_slabs: bad arguments");
  if (n % 4 == 0 && slab_stride % 4 == 0 && al16(slabs) && al16(out)) {
    const long long n4 = n / 4;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, slabs, out, n4, nslab, slab_stride);
  } else {
    hipLaunchKernelGGL(reduce_slabs_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, slabs, out, n, nslab, slab_stride);
  }
  DGVIT_CHECK_LAUNCH("reduce_slabs");
  return DGVIT_OK;
}
