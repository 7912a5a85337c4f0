// Token assembly around the patch-embedding GEMM (GoalFormer.py:138, 160-163) and small elementwise helpers.
//   patchify      : 'b (h p1) (w p2) -> b (h w) (p1 p2)'  (the GEMM's A operand; also kept for the weight gradient)
//   goal_row      : x0[b, 0, :] = goal[b] + pos[0]         (goal token in the CLS slot)
//   dropout       : in-place Bernoulli(keep) mask / keep, Philox4x32-10 keyed by (seed), counter = float4 index
//                   -- the same call with the same seed re-creates the mask in backward, nothing is stored
//   relu_bwd      : dpre = dy * (y > 0)                    (head MLPs)
#include "common.h"
#include "kernels.h"

namespace {

__global__ void __launch_bounds__(256) patchify_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int Hi, int Wi,
                                                       int ph, int pw) {
  const int gw = Wi / pw, gh = Hi / ph, pd = ph * pw;
  const long long total = (long long)B * gh * gw * pd;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int e = (int)(idx % pd);
  const long long bp = idx / pd;
  const int p = (int)(bp % (gh * gw));
  const long long b = bp / (gh * gw);
  const int p1 = e / pw, p2 = e % pw, hy = p / gw, wx = p % gw;
  out[idx] = img[(b * Hi + hy * ph + p1) * Wi + wx * pw + p2];
}

__global__ void __launch_bounds__(256) goal_row_kernel(const float* __restrict__ goal, const float* __restrict__ pos,
                                                       float* __restrict__ x0, int B, int N, int D) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)B * D) return;
  const int d = (int)(idx % D);
  const long long b = idx / D;
  x0[b * N * D + d] = goal[idx] + pos[d];
}

__global__ void __launch_bounds__(256) dropout_kernel(float* __restrict__ x, long long n4, unsigned long long seed,
                                                      const unsigned long long* __restrict__ seed_dev, float keep) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  if (seed_dev) seed = *seed_dev;   // graph-capturable form: the seed lives in device memory
  reinterpret_cast<float4*>(x)[i] = dropout4(reinterpret_cast<float4*>(x)[i], i, seed, keep);
}

__global__ void __launch_bounds__(256) relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out,
                                                       long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// out[r][:] = a[r][:] + b[r][:] over `rows` rows of `row4` float4 with independent row strides (in float4): the residual add of an
// Attention whose to_out is nn.Identity() (heads == 1 and dim_head == dim, GoalFormer.py:56,66-69,103)
__global__ void __launch_bounds__(256) add_rows_kernel(const float4* __restrict__ a, long long lda4, const float4* __restrict__ b, long long ldb4,
                                                       float4* __restrict__ out, long long ldo4, long long total4, int row4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const long long r = i / row4;
  const int c = (int)(i - r * row4);
  const float4 x = a[r * lda4 + c], y = b[r * ldb4 + c];
  out[r * ldo4 + c] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

// out[i][:] = src[idx[i]][:] for rows of `row4` float4 (device-resident replay sampling)
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ idx,
                                                          float* __restrict__ out, long long total4, int row4, long long nrows) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const long long r = i / row4;
  const int c = (int)(i % row4);
  long long s = idx[r];
  s = s < 0 ? 0 : (s >= nrows ? nrows - 1 : s);   // clamp instead of faulting on a bad index
  reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(src)[s * row4 + c];
}

}  // namespace

int gather_rows(const float* src, const long long* idx, float* out, long long nsel, long long row_floats, long long nrows,
                hipStream_t stream) {
  DGVIT_CHECK_ARG(src && idx && out && nsel > 0 && nrows > 0, "gather_rows: bad arguments");
  DGVIT_CHECK_ARG(row_floats > 0 && row_floats % 4 == 0 && row_floats / 4 < (1ll << 31), "gather_rows: row length must be a multiple of 4 floats");
  const long long total4 = nsel * (row_floats / 4);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, src, idx, out, total4,
                     (int)(row_floats / 4), nrows);
  DGVIT_CHECK_LAUNCH("gather_rows");
  return DGVIT_OK;
}

int add_rows(const float* a, long long lda, const float* b, long long ldb, float* out, long long ldo, long long rows, int cols, hipStream_t stream) {
  DGVIT_CHECK_ARG(a && b && out && rows > 0 && cols > 0, "add_rows: bad arguments");
  DGVIT_CHECK_ARG(cols % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0, "add_rows: row length and strides must be multiples of 4 floats");
  const long long total4 = rows * (cols / 4);
  hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const float4*>(a), lda / 4,
                     reinterpret_cast<const float4*>(b), ldb / 4, reinterpret_cast<float4*>(out), ldo / 4, total4, cols / 4);
  DGVIT_CHECK_LAUNCH("add_rows");
  return DGVIT_OK;
}

int patchify(const float* img, float* out, int B, int Hi, int Wi, int ph, int pw, hipStream_t stream) {
  DGVIT_CHECK_ARG(img && out && B > 0, "patchify: bad arguments");
  DGVIT_CHECK_ARG(ph > 0 && pw > 0 && Hi % ph == 0 && Wi % pw == 0, "Image dimensions must be divisible by the patch size.");
  const long long total = (long long)B * Hi * Wi;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, img, out, B, Hi, Wi, ph, pw);
  DGVIT_CHECK_LAUNCH("patchify");
  return DGVIT_OK;
}

int goal_row(const float* goal, const float* pos, float* x0, int B, int N, int D, hipStream_t stream) {
  DGVIT_CHECK_ARG(goal && pos && x0 && B > 0 && N > 0 && D > 0, "goal_row: bad arguments");
  const long long total = (long long)B * D;
  hipLaunchKernelGGL(goal_row_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, goal, pos, x0, B, N, D);
  DGVIT_CHECK_LAUNCH("goal_row");
  return DGVIT_OK;
}

int dropout_inplace(float* x, long long n, unsigned long long seed, const unsigned long long* seed_dev, float keep,
                    hipStream_t stream) {
  DGVIT_CHECK_ARG(x && n > 0 && n % 4 == 0 && keep > 0.f && keep <= 1.f, "dropout: bad arguments (n must be a multiple of 4)");
  const long long n4 = n / 4;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, n4, seed, seed_dev, keep);
  DGVIT_CHECK_LAUNCH("dropout");
  return DGVIT_OK;
}

int relu_bwd(const float* dy, const float* y, float* out, long long n, hipStream_t stream) {
  DGVIT_CHECK_ARG(dy && y && out && n > 0, "relu_bwd: bad arguments");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dy, y, out, n);
  DGVIT_CHECK_LAUNCH("relu_bwd");
  return DGVIT_OK;
}
