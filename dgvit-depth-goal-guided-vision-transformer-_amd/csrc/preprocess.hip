// depth-frame preprocessing in front of the encoder (env_lab.py:420-434 listener_callback, :78-89 add_nose,
// :69-76 blurring, :295-299 / :348-349 resize + /255), on device-resident float32 frames (B, H, W):
//   depth_normalize_u8 : cv2.normalize(NORM_MINMAX, 0, 255) per frame, then astype(uint8) (truncation; values kept as floats)
//   noise_clip         : clip(x + noise, 0, 255), noise given (parity tests) or drawn N(0, level) from Philox + Box-Muller
//   gaussian_blur      : cv2.GaussianBlur(ksize 5 or 11, sigma 0): separable, BORDER_REFLECT_101, on a band of rows
//   resize_bilinear    : cv2.resize(INTER_LINEAR) to the encoder's frame size, scaled (1/255)
// All of it is HBM-bound pixel work (a 440x640 frame is 1.1 MB); parity against OpenCV is UNPINNED (cv2 is not installed:
// the oracle restates its published formulas, tests/test_gpu_preprocess.py).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MM_BLOCKS = 64;   // partial min / max blocks per frame

__global__ void __launch_bounds__(256) minmax_partial_kernel(const float* __restrict__ src, float* __restrict__ part, long long n) {
  __shared__ float smin[4], smax[4];
  const float* f = src + (long long)blockIdx.y * n;
  float lo = INFINITY, hi = -INFINITY;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)MM_BLOCKS * 256) {
    const float v = f[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  lo = -wave_max(-lo);
  hi = wave_max(hi);
  if ((threadIdx.x & 63) == 0) {
    smin[threadIdx.x >> 6] = lo;
    smax[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + ((long long)blockIdx.y * MM_BLOCKS + blockIdx.x) * 2;
    o[0] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    o[1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  }
}

__global__ void __launch_bounds__(256) normalize_trunc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              const float* __restrict__ part, long long n) {
  float lo = INFINITY, hi = -INFINITY;
  const float* pp = part + (long long)blockIdx.y * MM_BLOCKS * 2;
  for (int i = 0; i < MM_BLOCKS; ++i) {   // 64 partials, read by every thread through the scalar cache
    lo = fminf(lo, pp[2 * i]);
    hi = fmaxf(hi, pp[2 * i + 1]);
  }
  const double d = (double)hi - (double)lo;
  const double scale = d > 2.220446049250313e-16 ? 255.0 / d : 0.0;       // cv2: scale 0 when max - min <= DBL_EPSILON
  const float a = (float)scale, b = (float)(-(double)lo * scale);          // scale and shift are formed in double, applied in float
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  // multiply and add rounded separately (no FMA contraction), as a scalar host implementation does: the truncation below makes
  // the last bit visible (the frame maximum lands on 255.0 exactly, not on 254.99998)
  if (i < n) dst[(long long)blockIdx.y * n + i] = truncf(__fadd_rn(__fmul_rn(src[(long long)blockIdx.y * n + i], a), b));
}

// (philox4x32_10: common.h)

// dst = clip(src + noise, 0, 255); noise == null: level * N(0, 1) from Philox4x32-10 (counter = float4 index) + Box-Muller
__global__ void __launch_bounds__(256) noise_clip_kernel(const float* __restrict__ src, const float* __restrict__ noise,
                                                         float* __restrict__ dst, long long n4, float level, unsigned long long seed) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 v = reinterpret_cast<const float4*>(src)[i], z;
  if (noise) {
    z = reinterpret_cast<const float4*>(noise)[i];
  } else {
    const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)(i >> 32), 0x66340000u, 0u), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const float u0 = (r.x + 0.5f) * 2.3283064365386963e-10f, u1 = r.y * 2.3283064365386963e-10f;
    const float u2 = (r.z + 0.5f) * 2.3283064365386963e-10f, u3 = r.w * 2.3283064365386963e-10f;
    const float ra = sqrtf(-2.f * logf(u0)) * level, rb = sqrtf(-2.f * logf(u2)) * level;
    z = make_float4(ra * cosf(6.283185307179586f * u1), ra * sinf(6.283185307179586f * u1), rb * cosf(6.283185307179586f * u3),
                    rb * sinf(6.283185307179586f * u3));
  }
  v.x = fminf(fmaxf(v.x + z.x, 0.f), 255.f);
  v.y = fminf(fmaxf(v.y + z.y, 0.f), 255.f);
  v.z = fminf(fmaxf(v.z + z.z, 0.f), 255.f);
  v.w = fminf(fmaxf(v.w + z.w, 0.f), 255.f);
  reinterpret_cast<float4*>(dst)[i] = v;
}

struct Taps { float k[11]; };

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i = (i < 0 ? -i : i) % p;
  return i >= n ? p - i : i;
}

// one pass of the separable blur on rows [y0, y1) of every frame: HORIZ ? along x : along y (reflection inside the band)
template <bool HORIZ>
__global__ void __launch_bounds__(256) blur_pass_kernel(const float* __restrict__ src, float* __restrict__ dst, Taps taps, int ksize, int H,
                                                        int W, int y0, int y1) {
  const int bh = y1 - y0;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)bh * W) return;
  const int y = (int)(idx / W), x = (int)(idx % W), r = ksize / 2;
  const float* f = src + ((long long)blockIdx.y * H + y0) * W;
  float s = 0.f;
  for (int t = 0; t < ksize; ++t) {
    const float v = HORIZ ? f[(long long)y * W + reflect101(x + t - r, W)] : f[(long long)reflect101(y + t - r, bh) * W + x];
    s += v * taps.k[t];
  }
  dst[((long long)blockIdx.y * H + y0 + y) * W + x] = s;
}

// cv2.resize(INTER_LINEAR): horizontal interpolation of the two source rows, then the vertical one; times `scale`
__global__ void __launch_bounds__(256) resize_bilinear_kernel(const float* __restrict__ src, float* __restrict__ dst, int Hs, int Ws,
                                                              int Hd, int Wd, float sy, float sx, float scale) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)Hd * Wd) return;
  const int y = (int)(idx / Wd), x = (int)(idx % Wd);
  float fx = (x + 0.5f) * sx - 0.5f, fy = (y + 0.5f) * sy - 0.5f;
  int x0 = (int)floorf(fx), yy0 = (int)floorf(fy);
  fx -= x0;
  fy -= yy0;
  if (x0 < 0) { x0 = 0; fx = 0.f; }
  if (x0 >= Ws - 1) { x0 = Ws - 1; fx = 0.f; }
  if (yy0 < 0) { yy0 = 0; fy = 0.f; }
  if (yy0 >= Hs - 1) { yy0 = Hs - 1; fy = 0.f; }
  const int x1 = min(x0 + 1, Ws - 1), yy1 = min(yy0 + 1, Hs - 1);
  const float* f = src + (long long)blockIdx.y * Hs * Ws;
  const float top = f[(long long)yy0 * Ws + x0] * (1.f - fx) + f[(long long)yy0 * Ws + x1] * fx;
  const float bot = f[(long long)yy1 * Ws + x0] * (1.f - fx) + f[(long long)yy1 * Ws + x1] * fx;
  dst[(long long)blockIdx.y * Hd * Wd + idx] = (top * (1.f - fy) + bot * fy) * scale;
}

Taps make_taps(int ksize) {
  Taps t = {};
  if (ksize == 5) {   // OpenCV's fixed small kernel for ksize 5, sigma <= 0
    const float k5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    for (int i = 0; i < 5; ++i) t.k[i] = k5[i];
    return t;
  }
  const double sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8;
  double k[11], sum = 0;
  for (int i = 0; i < ksize; ++i) {
    const double x = i - (ksize - 1) * 0.5;
    k[i] = exp(-(x * x) / (2.0 * sigma * sigma));
    sum += k[i];
  }
  for (int i = 0; i < ksize; ++i) t.k[i] = (float)(k[i] / sum);
  return t;
}

}  // namespace

int depth_normalize_u8(const float* src, float* dst, float* part, int B, int H, int W, hipStream_t st) {
  DGVIT_CHECK_ARG(src && dst && part && B > 0 && H > 0 && W > 0, "depth_normalize_u8: bad arguments");
  const long long n = (long long)H * W;
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(MM_BLOCKS, B), dim3(256), 0, st, src, part, n);
  hipLaunchKernelGGL(normalize_trunc_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, src, dst, part, n);
  DGVIT_CHECK_LAUNCH("depth_normalize_u8");
  return DGVIT_OK;
}
long long depth_normalize_scratch_floats(int B) { return (long long)B * MM_BLOCKS * 2; }

int noise_clip(const float* src, const float* noise, float* dst, long long n, float level, unsigned long long seed, hipStream_t st) {
  DGVIT_CHECK_ARG(src && dst && n > 0 && n % 4 == 0, "noise_clip: n must be a positive multiple of 4");
  hipLaunchKernelGGL(noise_clip_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, src, noise, dst, n / 4, level, seed);
  DGVIT_CHECK_LAUNCH("noise_clip");
  return DGVIT_OK;
}

// dst rows [y0, y1) = blur of src rows [y0, y1) (other rows of dst untouched); tmp: B*H*W floats; src may equal dst
int gaussian_blur_band(const float* src, float* dst, float* tmp, int B, int H, int W, int ksize, int y0, int y1, hipStream_t st) {
  DGVIT_CHECK_ARG(src && dst && tmp && B > 0 && H > 0 && W > 0, "gaussian_blur: bad arguments");
  DGVIT_CHECK_ARG(ksize == 5 || ksize == 11, "gaussian_blur: ksize %d unsupported (5 or 11)", ksize);
  DGVIT_CHECK_ARG(y0 >= 0 && y1 <= H, "gaussian_blur: bad row band");
  if (y1 <= y0) return DGVIT_OK;
  const Taps t = make_taps(ksize);
  const long long n = (long long)(y1 - y0) * W;
  const dim3 grid((unsigned)((n + 255) / 256), B);
  hipLaunchKernelGGL((blur_pass_kernel<true>), grid, dim3(256), 0, st, src, tmp, t, ksize, H, W, y0, y1);
  hipLaunchKernelGGL((blur_pass_kernel<false>), grid, dim3(256), 0, st, (const float*)tmp, dst, t, ksize, H, W, y0, y1);
  DGVIT_CHECK_LAUNCH("gaussian_blur");
  return DGVIT_OK;
}

int resize_bilinear(const float* src, float* dst, int B, int Hs, int Ws, int Hd, int Wd, float scale, hipStream_t st) {
  DGVIT_CHECK_ARG(src && dst && B > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "resize_bilinear: bad arguments");
  const long long n = (long long)Hd * Wd;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, src, dst, Hs, Ws, Hd, Wd,
                     (float)Hs / (float)Hd, (float)Ws / (float)Wd, scale);
  DGVIT_CHECK_LAUNCH("resize_bilinear");
  return DGVIT_OK;
}
