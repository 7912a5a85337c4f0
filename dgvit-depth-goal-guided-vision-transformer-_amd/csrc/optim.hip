// Optimiser-side kernels for the step AFTER the hot path: the reference runs
// torch.optim.Adam over ~70 tensors per network (DRL.py:401-403,412-414) and a per-parameter Python loop for the
// Polyak target update (utils.py:31-33).  Here both are one HBM-bound pass over a flat fp32 buffer.
//   adam_step   : torch.optim.Adam semantics (bias-corrected, eps added to sqrt(v_hat), optional L2 weight decay)
//   soft_update : target <- target * (1 - tau) + source * tau
// Algorithmic bytes: Adam 28 B/parameter (read p,g,m,v; write p,m,v), soft update 12 B/parameter.
#include "common.h"
#include "kernels.h"

namespace {

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n4, float lr_c, float beta1, float beta2,
                                                   float inv_sqrt_bc2, float eps, float wd, float lr,
                                                   const long long* __restrict__ step_dev) {
  if (step_dev) {   // graph-capturable form: bias corrections from the device-side step counter
    const float t = (float)*step_dev;
    lr_c = lr / (1.f - powf(beta1, t));
    inv_sqrt_bc2 = rsqrtf(1.f - powf(beta2, t));
  }
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv0 = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv0.x, gv0.y, gv0.z, gv0.w};
    float me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = ge[e] + wd * pe[e];
      me[e] = beta1 * me[e] + (1.f - beta1) * gg;
      ve[e] = beta2 * ve[e] + (1.f - beta2) * gg * gg;
      pe[e] -= lr_c * me[e] / (sqrtf(ve[e]) * inv_sqrt_bc2 + eps);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(me[0], me[1], me[2], me[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
  }
}

__global__ void __launch_bounds__(256) soft_update_kernel(float* __restrict__ tgt, const float* __restrict__ src, long long n4,
                                                          float tau) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 t = reinterpret_cast<float4*>(tgt)[i];
    const float4 s = reinterpret_cast<const float4*>(src)[i];
    t.x = t.x * (1.f - tau) + s.x * tau;
    t.y = t.y * (1.f - tau) + s.y * tau;
    t.z = t.z * (1.f - tau) + s.z * tau;
    t.w = t.w * (1.f - tau) + s.w * tau;
    reinterpret_cast<float4*>(tgt)[i] = t;
  }
}

inline unsigned grid_for(long long n4) {
  long long b = (n4 + 255) / 256;
  if (b > 2048) b = 2048;  // 8 workgroups per CU, grid-stride beyond
  return (unsigned)(b < 1 ? 1 : b);
}
inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// n must be a multiple of 4 and the buffers 16-byte aligned (the flat buffers of dgvit_amd.optim are)
int adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
              float weight_decay, long long step, const long long* step_dev, hipStream_t stream) {
  DGVIT_CHECK_ARG(p && g && m && v && n > 0 && n % 4 == 0, "adam_step: n must be a positive multiple of 4");
  DGVIT_CHECK_ARG(al16(p) && al16(g) && al16(m) && al16(v), "adam_step: buffers must be 16-byte aligned");
  DGVIT_CHECK_ARG((step >= 1 || step_dev) && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "adam_step: bad hyper-parameters");
  const double bc1 = 1.0 - pow((double)beta1, (double)(step >= 1 ? step : 1));
  const double bc2 = 1.0 - pow((double)beta2, (double)(step >= 1 ? step : 1));
  const int slot = profile_begin(PROF_OTHER, 0.0, stream);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, p, g, m, v, n / 4, (float)(lr / bc1), beta1, beta2,
                     (float)(1.0 / sqrt(bc2)), eps, weight_decay, lr, step_dev);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("adam_step");
  return DGVIT_OK;
}

int soft_update(float* target, const float* source, long long n, float tau, hipStream_t stream) {
  DGVIT_CHECK_ARG(target && source && n > 0 && n % 4 == 0, "soft_update: n must be a positive multiple of 4");
  DGVIT_CHECK_ARG(al16(target) && al16(source), "soft_update: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(soft_update_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, target, source, n / 4, tau);
  DGVIT_CHECK_LAUNCH("soft_update");
  return DGVIT_OK;
}
