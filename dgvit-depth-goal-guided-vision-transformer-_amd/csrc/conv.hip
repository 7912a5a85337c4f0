// CNN feature stack of the reference's shipped critic / CNN actor:
//   Conv2d(1,16,5,s2) -> ReLU -> Conv2d(16,64,5,s2) -> ReLU -> Conv2d(64,256,5,s2) -> ReLU -> AdaptiveAvgPool2d(1)
// (got_sac_network.py:129-133,151-155 QNetwork; :263-266,292-296 GaussianPolicy), forward and backward.
//
// Each convolution is an implicit GEMM made explicit: activations are kept NHWC, `im2col` lays the 5x5xC
// receptive fields out as rows [(b,oh,ow)][(kh,kw,c)] and the fp32 MFMA GEMM (gemm.hip) multiplies them with the
// weight permuted to [cout][(kh,kw,cin)], bias + ReLU fused in its epilogue; the output rows ARE the next layer's
// NHWC input.  Backward: dW = dY^T cols (split-K GEMM, bias gradient fused), dcols = dY W (GEMM), and a gather-style
// col2im (each input pixel sums the <= 9 windows that cover it: deterministic, no atomics) with the previous
// layer's ReLU mask fused.  The single-channel first layer pads K = 25 to 28 so that it takes the float4 path.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int KS = 5, STRIDE = 2;

// cols[(b,oh,ow)][(kh*KS + kw)*C + c] = x[b][oh*2+kh][ow*2+kw][c]; C % 4 == 0, one float4 per thread
__global__ void __launch_bounds__(256) im2col_c4_kernel(const float* __restrict__ x, float* __restrict__ cols, int B, int H, int W, int C,
                                                        int OH, int OW) {
  const int c4n = C / 4;
  const long long total = (long long)B * OH * OW * KS * KS * c4n;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % c4n);
  long long r = idx / c4n;
  const int kw = (int)(r % KS); r /= KS;
  const int kh = (int)(r % KS); r /= KS;
  const int ow = (int)(r % OW); r /= OW;
  const int oh = (int)(r % OH);
  const long long b = r / OH;
  const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + oh * STRIDE + kh) * W + ow * STRIDE + kw) * C + c4 * 4);
  reinterpret_cast<float4*>(cols)[idx] = v;
}

// single-channel input: row of KP (>= 25, multiple of 4) floats, tail zero
__global__ void __launch_bounds__(256) im2col_c1_kernel(const float* __restrict__ x, float* __restrict__ cols, int B, int H, int W, int OH,
                                                        int OW, int KP) {
  const long long total = (long long)B * OH * OW * KP;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx % KP);
  long long r = idx / KP;
  const int ow = (int)(r % OW); r /= OW;
  const int oh = (int)(r % OH);
  const long long b = r / OH;
  float v = 0.f;
  if (k < KS * KS) v = x[(b * H + oh * STRIDE + k / KS) * W + ow * STRIDE + k % KS];
  cols[idx] = v;
}

// dx[b][y][x][c] = relu'(xin) * sum over windows (oh,ow,kh,kw) with oh*2+kh == y, ow*2+kw == x of dcols[...]
__global__ void __launch_bounds__(256) col2im_relu_kernel(const float* __restrict__ dcols, const float* __restrict__ xin,
                                                          float* __restrict__ dx, int B, int H, int W, int C, int OH, int OW) {
  const int c4n = C / 4;
  const long long total = (long long)B * H * W * c4n;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % c4n);
  long long r = idx / c4n;
  const int xw = (int)(r % W); r /= W;
  const int y = (int)(r % H);
  const long long b = r / H;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long rowlen = (long long)KS * KS * C;
  for (int kh = y & 1; kh < KS; kh += 2) {
    const int oh = (y - kh) / 2;
    if (y - kh < 0 || oh >= OH) continue;
    for (int kw = xw & 1; kw < KS; kw += 2) {
      const int ow = (xw - kw) / 2;
      if (xw - kw < 0 || ow >= OW) continue;
      const float4 v = *reinterpret_cast<const float4*>(dcols + ((b * OH + oh) * OW + ow) * rowlen + (kh * KS + kw) * C + c4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  const float4 m = reinterpret_cast<const float4*>(xin)[idx];
  s.x = m.x > 0.f ? s.x : 0.f; s.y = m.y > 0.f ? s.y : 0.f; s.z = m.z > 0.f ? s.z : 0.f; s.w = m.w > 0.f ? s.w : 0.f;
  reinterpret_cast<float4*>(dx)[idx] = s;
}

// reference layout (cout, cin, KS, KS) <-> packed (cout, KP) with k = (kh*KS + kw)*cin + c; KP >= KS*KS*cin, tail zero.
// unpack == 0: dst (packed) <- src (reference);  unpack == 1: dst (reference) <- src (packed)
__global__ void __launch_bounds__(256) weight_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin, int KP,
                                                          int unpack) {
  const long long total = (long long)cout * KP;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx % KP);
  const int o = (int)(idx / KP);
  if (k >= KS * KS * cin) {
    if (!unpack) dst[idx] = 0.f;
    return;
  }
  const int c = k % cin, kk = k / cin;
  const long long ref = ((long long)o * cin + c) * KS * KS + kk;
  if (unpack) dst[ref] = src[idx];
  else dst[idx] = src[ref];
}

// feat[b][c] = mean over S rows of x[(b*S + s)][c]      (AdaptiveAvgPool2d(1) on NHWC rows; token mean of GoT pool='mean')
// 256 threads = 64 channels x 4 row lanes, 4 independent loads in flight per thread, fixed-order LDS combine.
__global__ void __launch_bounds__(256) avgpool_kernel(const float* __restrict__ x, float* __restrict__ feat, int S, int C) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int b = blockIdx.x, c = blockIdx.y * 64 + tx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    const float* base = x + (long long)b * S * C + c;
    int i = ty;
    for (; i + 12 < S; i += 16) {
      s0 += base[(long long)i * C];
      s1 += base[(long long)(i + 4) * C];
      s2 += base[(long long)(i + 8) * C];
      s3 += base[(long long)(i + 12) * C];
    }
    for (; i < S; i += 4) s0 += base[(long long)i * C];
  }
  part[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && c < C) feat[(long long)b * C + c] = (((part[0][tx] + part[1][tx]) + part[2][tx]) + part[3][tx]) / (float)S;
}

// dy[(b*S+s)][c] = x > 0 ? dfeat[b][c] / S : 0
__global__ void __launch_bounds__(256) avgpool_bwd_relu_kernel(const float* __restrict__ dfeat, const float* __restrict__ x,
                                                               float* __restrict__ dy, long long total4, int S, int C) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int c4n = C / 4;
  const int c4 = (int)(idx % c4n);
  const long long b = idx / c4n / S;
  const float4 g = *reinterpret_cast<const float4*>(dfeat + b * C + c4 * 4);
  const float4 m = reinterpret_cast<const float4*>(x)[idx];
  const float inv = 1.f / (float)S;
  reinterpret_cast<float4*>(dy)[idx] = make_float4(m.x > 0.f ? g.x * inv : 0.f, m.y > 0.f ? g.y * inv : 0.f,
                                                   m.z > 0.f ? g.z * inv : 0.f, m.w > 0.f ? g.w * inv : 0.f);
}

// dx[(b*S+s)][c] = dfeat[b][c] / S      (gradient of a mean over S rows)
__global__ void __launch_bounds__(256) mean_bwd_kernel(const float* __restrict__ dfeat, float* __restrict__ dx, long long total, int S,
                                                       int C) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long long b = idx / C / S;
  dx[idx] = dfeat[b * C + c] / (float)S;
}

inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

int im2col(const float* x, float* cols, int B, int H, int W, int C, int OH, int OW, int KP, hipStream_t st) {
  if (C == 1) {
    hipLaunchKernelGGL(im2col_c1_kernel, dim3(nblk((long long)B * OH * OW * KP)), dim3(256), 0, st, x, cols, B, H, W, OH, OW, KP);
  } else {
    DGVIT_CHECK_ARG(C % 4 == 0 && KP == KS * KS * C, "im2col: channels must be 1 or a multiple of 4");
    hipLaunchKernelGGL(im2col_c4_kernel, dim3(nblk((long long)B * OH * OW * KS * KS * (C / 4))), dim3(256), 0, st, x, cols, B, H, W,
                       C, OH, OW);
  }
  DGVIT_CHECK_LAUNCH("im2col");
  return DGVIT_OK;
}

int col2im_relu(const float* dcols, const float* xin, float* dx, int B, int H, int W, int C, int OH, int OW, hipStream_t st) {
  DGVIT_CHECK_ARG(C % 4 == 0, "col2im: channels must be a multiple of 4");
  hipLaunchKernelGGL(col2im_relu_kernel, dim3(nblk((long long)B * H * W * (C / 4))), dim3(256), 0, st, dcols, xin, dx, B, H, W, C, OH, OW);
  DGVIT_CHECK_LAUNCH("col2im");
  return DGVIT_OK;
}

int weight_pack(const float* src, float* dst, int cout, int cin, int KP, int unpack, hipStream_t st) {
  hipLaunchKernelGGL(weight_pack_kernel, dim3(nblk((long long)cout * KP)), dim3(256), 0, st, src, dst, cout, cin, KP, unpack);
  DGVIT_CHECK_LAUNCH("weight_pack");
  return DGVIT_OK;
}

int avgpool(const float* x, float* feat, int B, int S, int C, hipStream_t st) {
  hipLaunchKernelGGL(avgpool_kernel, dim3(B, (C + 63) / 64), dim3(256), 0, st, x, feat, S, C);
  DGVIT_CHECK_LAUNCH("avgpool");
  return DGVIT_OK;
}

int avgpool_bwd_relu(const float* dfeat, const float* x, float* dy, int B, int S, int C, hipStream_t st) {
  DGVIT_CHECK_ARG(C % 4 == 0, "avgpool_bwd: channels must be a multiple of 4");
  const long long total4 = (long long)B * S * (C / 4);
  hipLaunchKernelGGL(avgpool_bwd_relu_kernel, dim3(nblk(total4)), dim3(256), 0, st, dfeat, x, dy, total4, S, C);
  DGVIT_CHECK_LAUNCH("avgpool_bwd");
  return DGVIT_OK;
}

int mean_bwd(const float* dfeat, float* dx, int B, int S, int C, hipStream_t st) {
  const long long total = (long long)B * S * C;
  hipLaunchKernelGGL(mean_bwd_kernel, dim3(nblk(total)), dim3(256), 0, st, dfeat, dx, total, S, C);
  DGVIT_CHECK_LAUNCH("mean_bwd");
  return DGVIT_OK;
}
