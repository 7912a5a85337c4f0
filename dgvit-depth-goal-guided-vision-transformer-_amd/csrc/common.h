// Shared declarations for the DGViT gfx950 kernels (internal; the public C ABI is include/dgvit_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/dgvit_hip.h"
#include "knobs.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error reporting (thread-local message, negative return codes from include/dgvit_hip.h) ------
int dgvit_set_error(int code, const char* fmt, ...);

#define DGVIT_CHECK_ARG(cond, ...)                                   \
  do {                                                               \
    if (!(cond)) return dgvit_set_error(DGVIT_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define DGVIT_CHECK_LAUNCH(name)                                                                 \
  do {                                                                                           \
    hipError_t e_ = hipGetLastError();                                                           \
    if (e_ != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

// Kernel attributes (the dynamic-LDS limit above 64 KB) are per DEVICE: a flag per device ordinal, set on the first launch on
// that device (a process-wide `static bool` would leave a second GPU of a single-process host without the attribute).
// Racing threads may both set the attribute: harmless, it is idempotent.
struct DeviceOnce {
  std::atomic<unsigned long long> done{0};
  unsigned long long pending() const {    // 0: already done on the current device, else the device's bit for mark()
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    return (done.load(std::memory_order_relaxed) & bit) ? 0ull : bit;
  }
  void mark(unsigned long long bit) { done.fetch_or(bit, std::memory_order_relaxed); }
};

// ---- GEMM -----------------------------------------------------------------------------------
// C[m][n] = sum_k Aop[m][k] * Bop[k][n]   (fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32)
enum GemmLayout {
  GEMM_NT = 0,  // A [M][K] (k contiguous), B [N][K] (k contiguous)   forward:  Y = X W^T
  GEMM_NN = 1,  // A [M][K] (k contiguous), B [K][N] (n contiguous)   dgrad:    dX = dY W
  GEMM_TN = 2   // A [K][M] (m contiguous), B [K][N] (n contiguous)   wgrad:    dW = dY^T X
};

enum GemmEpilogue {
  EPI_STORE = 0,   // C = acc (+ bias[n]) (+ res[rr][n]),  optional output-row remap
  EPI_GELU2 = 1,   // C = acc + bias ; C2 = gelu_erf(C)
  EPI_DGELU = 2,   // C = acc * gelu'(aux[m][n])
  EPI_RELU = 3,    // C = max(acc + bias, 0)
  EPI_DRELU = 4,   // C = aux[m][n] > 0 ? acc : 0
  EPI_SPLITK = 5,  // C = slab[z][m][n] = acc   (reduced later by dgvit_reduce_slabs)
  EPI_GELU2D = 6,  // t = acc + bias ; C = gelu_erf'(t) ; C2 = gelu_erf(t)        (training forward of fc1: the factor the backward needs)
  EPI_DMUL = 7,    // C = acc * aux[m][n]                                          (its data gradient: dh = (dy W2) * gelu'(h), factor stored)
  EPI_GELU = 8     // C = gelu_erf(acc + bias)                                     (no-grad forward of fc1: the pre-activation is never stored)
};

struct GemmParams {
  const float* A; const float* B;
  int lda, ldb;
  int M, N, K;
  int kchunk;            // K range handled by one blockIdx.z (multiple of BK); == K when not split
  int a_kgrp;            // must be 0
  // patch gather of the A operand (NT, patch embedding; GoalFormer.py:138 'b (h p1) (w p2) -> b (h w) (p1 p2)' folded into the
  // tile loader): g_img != null => A[m][k] = img[b][hy * ph + p1][wx * pw + p2] with m = b * P + hy * gw + wx, k = p1 * pw + p2;
  // `A` / `lda` are ignored.  Needs pw % 4 == 0, image width % 4 == 0 and a 16-byte aligned image (float4 loads never cross a patch row).
  const float* g_img; long long g_img_floats;
  int g_wi, g_hw, g_ph, g_pw, g_gw, g_P, g_inv;   // image width, image height * width, patch h / w, patches per row / frame, 65536 / pw + 1
  // the same loader reads strided, overlapping windows (NHWC convolution, C % 4 == 0: g_pw = KW * C window-row floats, g_wi = W * C,
  // g_hw = H * W * C): g_kh = window rows (0: g_ph), g_ph = row step, g_xs = window step along a row in floats (0: g_pw),
  // g_shift = the multiply-shift of k / g_pw (0: 16; g_inv = 2^g_shift / g_pw + 1)
  int g_kh, g_xs, g_shift;
  float* C; int ldc;
  const float* bias;     // [N] or null
  const float* res; int ldr;  // residual [*][N] or null
  int res_mod;           // > 0: residual row = (m % res_mod) + 1  (positional embedding broadcast over frames)
  float* C2; int ldc2;   // second output (EPI_GELU2)
  const float* aux; int ldaux;  // EPI_DGELU / EPI_DRELU input
  int c_rgrp;            // > 0: physical C row = m + m / c_rgrp + 1
  long long slab_stride; // EPI_SPLITK: floats between consecutive z slabs
  int colsum;            // EPI_SPLITK: also write sum_k A[k][m] at slab[z][M*N + m] (bias gradient)
  int evec;              // set by gemm_f32: epilogue may use float4 global accesses
  // in-launch split-K of the NT / NN forms (gemm_split_plan): tiles >= split_from are cut into nsplit k-slices of kchunk_split;
  // `counters` (one int per tile, zero on entry, left zero) and `slabs` (gemm_split_plan().slab_floats) come from the caller;
  // counters == nullptr disables splitting
  int* counters; float* slabs;
  long long slab_capacity; int counter_capacity;   // floats / ints available behind `slabs` / `counters`
  int split_from, nsplit, kchunk_split;            // filled in at launch
  int zsplit;                                      // EPI_SPLITK: k-slices of the launch when they are folded into blockIdx.x (0: blockIdx.z)
  // fused LayerNorm of the output rows (EPI_STORE, N == the tile's width: the whole row is in one tile; the encoder uses it for
  // D <= 64): y[m] = LN(C[m]) * g + b with row stride ln_ld, mean / rstd per logical row; ln_y == nullptr: off
  const float* ln_g; const float* ln_b; float* ln_y; float* ln_mean; float* ln_rstd; long long ln_ld; float ln_eps;
  long long* stamps; int stamp_capacity;           // diagnostic (dgvit_set_gemm_stamps): 16 counters per workgroup, or null
  int diag;                                        // A/B knob (dgvit_set_gemm_diagnostics)
};

struct GemmSplitPlan {
  int tiles, split_from, nsplit, kchunk;   // nsplit <= 1: no split
  long long slab_floats;                   // scratch the launch needs
};
// the split decision for C (M x N) = A B over K with the tile the automatic choice picks; used for sizing and at launch
GemmSplitPlan gemm_split_plan(int layout, int M, int N, int K);
GemmSplitPlan gemm_split_plan_gather(int M, int N, int K);   // A gathered from an image: the 64 x 64 x 32 tile

// live timing hooks (profile.hip); slot < 0 = not recording
int profile_begin(int kind, double work, hipStream_t st);
void profile_end(int slot, hipStream_t st);
enum { PROF_GEMM = 0, PROF_ATTN_FWD = 1, PROF_ATTN_BWD = 2, PROF_OTHER = 3 };

int gemm_f32(int layout, int epi, const GemmParams& p, int nsplit, hipStream_t stream);

// Deterministic fixed-order reductions of split-K slabs / per-workgroup column partials, several of them per launch.
#define DGVIT_REDUCE_JOBS 8
struct ReduceJob {
  const float* slabs; float* out1; float* out2;
  long long n4, n14, stride4;   // float4 columns in all / going to out1; float4s between consecutive slabs
  int nslab, cw_log;            // slabs to sum; log2 of the float4 columns per 256-thread block (6 or 4)
};
struct ReduceGroup {
  ReduceJob job[DGVIT_REDUCE_JOBS];
  int first_block[DGVIT_REDUCE_JOBS + 1];
  int njobs;
};
void reduce_group_init(ReduceGroup& g);
int reduce_group_add(ReduceGroup& g, const float* slabs, float* out1, long long n1, float* out2, long long n, int nslab,
                     long long slab_stride, hipStream_t stream);
int reduce_group_flush(ReduceGroup& g, hipStream_t stream);
int reduce_slabs(const float* slabs, float* out, long long n, int nslab, long long slab_stride, hipStream_t stream);
int reduce_slabs2(const float* slabs, float* out1, long long n1, float* out2, long long n, int nslab, long long slab_stride,
                  hipStream_t stream);

// erf with |abs error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26) on v_rcp_f32 / v_exp_f32: ~13 VALU instructions
// instead of ~37 for ocml's erff (5.5e-7 once evaluated in fp32).  GELU(x) and GELU'(x) built on it stay within ~5e-7 of the exact-erf forms for
// |x| <= 6 (checked against fp64 in tests/test_gpu_ops.py::test_gemm_epilogues), far inside the 1e-4 parity gate.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(fmaf(-p, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  return fmaf(x, pdf, cdf);
}

// gelu and its derivative together: the erf form's exp(-x^2 / 2) is the density's, so the derivative costs two more FMAs.  The two
// results are bit-identical to gelu_erf / gelu_erf_grad (same operations in the same order).
__device__ __forceinline__ void gelu_erf_both(float x, float& gelu, float& grad) {
  gelu = gelu_erf(x);
  grad = gelu_erf_grad(x);
}

// ---- train-mode nn.Dropout (GoalFormer.py:144,163): Bernoulli(keep) mask / keep from Philox4x32-10 keyed by the seed, counter = the
// float4 index of the element group in the (T, D) tensor.  One definition for the stand-alone kernel (embed.hip) and the kernels that
// apply the mask while they stage rows (block.hip): the same seed gives the same mask wherever it is applied.
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0;
    key.y += W1;
  }
  return ctr;
}

__device__ __forceinline__ float4 dropout4(float4 v, long long i4, unsigned long long seed, float keep) {
  const uint4 r = philox4x32_10(make_uint4((uint32_t)i4, (uint32_t)(i4 >> 32), 0u, 0u), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  const float inv = 1.0f / keep;
  const float sc = 2.3283064365386963e-10f;  // 2^-32
  v.x = (r.x * sc < keep) ? v.x * inv : 0.f;
  v.y = (r.y * sc < keep) ? v.y * inv : 0.f;
  v.z = (r.z * sc < keep) ? v.z * inv : 0.f;
  v.w = (r.w * sc < keep) ? v.w * inv : 0.f;
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
