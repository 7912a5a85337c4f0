// Row normalisations and column reductions of the DGViT encoder (HBM-bound; one wave64 per token row).
//   LayerNorm  fwd/bwd : PreNorm's nn.LayerNorm, eps 1e-5 (GoalFormer.py:34,37)
//   RMSNorm    fwd/bwd : F.normalize(x) * sqrt(D) * g, eps 1e-12 (GoalFormer.py:120-122)
//   colsum             : bias gradients  db[n] = sum_t dY[t][n]
// Column reductions (dgamma/dbeta/dg/db) are two-stage and deterministic: every workgroup writes one
// partial row into a slab, reduce_slabs() adds the slabs in a fixed order.
#include "common.h"
#include "kernels.h"
#include "bf16.h"

namespace {

__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
  const fx4 v = __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), fx4);
  return make_float4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------- LayerNorm forward
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int T, int D,
                                                            float eps, int rs) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const float* xr = x + (long long)row * rs * D;   // logical row r lives at physical row r*rs (rs = N: token 0 of every frame)
  float s = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    s += (v.x + v.y) + (v.z + v.w);
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    const float a = v.x - mu, b = v.y - mu, cc = v.z - mu, d = v.w - mu;
    q += (a * a + b * b) + (cc * cc + d * d);
  }
  const float rsd = rsqrtf(wave_sum(q) / (float)D + eps);
  float* yr = y + (long long)row * rs * D;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 b = *reinterpret_cast<const float4*>(beta + c);
    float4 o;
    o.x = (v.x - mu) * rsd * g.x + b.x;
    o.y = (v.y - mu) * rsd * g.y + b.y;
    o.z = (v.z - mu) * rsd * g.z + b.z;
    o.w = (v.w - mu) * rsd * g.w + b.w;
    *reinterpret_cast<float4*>(yr + c) = o;
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rsd;
  }
}

// ---------------------------------------------------------------- LayerNorm backward
// dx[t] = dres[t] + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// partial[blk][0][c] = sum_rows dy * xhat (dgamma),  partial[blk][1][c] = sum_rows dy (dbeta)
// DYT = float (fp32 path) or bf16_t (bf16 configuration: dy is a bf16 GEMM output; dxb, if given, receives a bf16 copy of
// dx -- the operand of the next data- and weight-gradient GEMMs)
template <int NCH, typename DYT>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const DYT* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ dres,
                                                            float* __restrict__ dx, bf16_t* __restrict__ dxb,
                                                            float* __restrict__ partial, int T, int D, int rs) {
  __shared__ float red[4][2][NCH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 ag[NCH], ab[NCH], gm[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = ag[i];
    const int c = lane * 4 + i * 256;
    gm[i] = c < D ? *reinterpret_cast<const float4*>(gamma + c) : ag[i];
  }
  const float invD = 1.f / (float)D;
  for (int row = blockIdx.x * 4 + wave; row < T; row += gridDim.x * 4) {
    const long long off = (long long)row * rs * D;
    const float mu = mean[row], rsd = rstd[row];
    float4 g[NCH], xh[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane * 4 + i * 256;
      g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      xh[i] = g[i];
      if (c < D) {
        const float4 d = load4(dy + off + c);
        const float4 v = *reinterpret_cast<const float4*>(x + off + c);
        xh[i] = make_float4((v.x - mu) * rsd, (v.y - mu) * rsd, (v.z - mu) * rsd, (v.w - mu) * rsd);
        ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
        g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      }
    }
    const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane * 4 + i * 256;
      if (c < D) {
        float4 o;
        o.x = rsd * (g[i].x - c1 - xh[i].x * c2);
        o.y = rsd * (g[i].y - c1 - xh[i].y * c2);
        o.z = rsd * (g[i].z - c1 - xh[i].z * c2);
        o.w = rsd * (g[i].w - c1 - xh[i].w * c2);
        if (dres) {
          const float4 r = *reinterpret_cast<const float4*>(dres + off + c);
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        *reinterpret_cast<float4*>(dx + off + c) = o;
        if (dxb) *reinterpret_cast<bf16x4*>(dxb + off + c) = __builtin_convertvector(fx4{o.x, o.y, o.z, o.w}, bf16x4);
      }
    }
  }
  // block reduction of the per-wave column partials (fixed wave order -> deterministic)
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    *reinterpret_cast<float4*>(&red[wave][0][i * 256 + lane * 4]) = ag[i];
    *reinterpret_cast<float4*>(&red[wave][1][i * 256 + lane * 4]) = ab[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * NCH * 256; c += 256) {
    const int which = c / (NCH * 256), col = c % (NCH * 256);
    if (col < D) {
      const float s = ((red[0][which][col] + red[1][which][col]) + red[2][which][col]) + red[3][which][col];
      partial[((long long)blockIdx.x * 2 + which) * D + col] = s;
    }
  }
}

// ---------------------------------------------------------------- RMSNorm (one wave per frame row)
__global__ void __launch_bounds__(256) rmsnorm_fwd_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ g,
                                                          float* __restrict__ y, int B, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* xr = x + row * ldx;
  float q = 0.f;
  for (int c = lane; c < D; c += 64) q += xr[c] * xr[c];
  const float n = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
  const float sc = sqrtf((float)D);
  for (int c = lane; c < D; c += 64) y[(long long)row * D + c] = xr[c] / n * sc * g[c];
}

// dx = u/n - x * (u.x)/n^3 with u = dy*g*sqrt(D)  (second term dropped when the norm was clamped);
// partial[blk][c] = sum_rows dy * x/n * sqrt(D)
__global__ void __launch_bounds__(256) rmsnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, long long ldx,
                                                          const float* __restrict__ g, float* __restrict__ dx, long long lddx,
                                                          float* __restrict__ partial, int B, int D) {
  extern __shared__ float red[];  // [4][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float sc = sqrtf((float)D);
  for (int c = lane; c < D; c += 64) red[wave * D + c] = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < B; row += gridDim.x * 4) {
    const float* xr = x + row * ldx;
    const float* dr = dy + (long long)row * D;
    float q = 0.f, ux = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float xv = xr[c];
      q += xv * xv;
      ux += dr[c] * g[c] * sc * xv;
    }
    q = wave_sum(q);
    ux = wave_sum(ux);
    const float nr = sqrtf(q);
    const bool clamped = nr < 1e-12f;
    const float n = clamped ? 1e-12f : nr;
    const float k = clamped ? 0.f : ux / (n * n * n);
    for (int c = lane; c < D; c += 64) {
      const float xv = xr[c], dv = dr[c];
      dx[row * lddx + c] = dv * g[c] * sc / n - xv * k;
      red[wave * D + c] += dv * xv / n * sc;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256)
    partial[(long long)blockIdx.x * D + c] = ((red[c] + red[D + c]) + red[2 * D + c]) + red[3 * D + c];
}

// ---------------------------------------------------------------- column sums
// grid (ceil(N/64), RB); block 256 = 64 columns x 4 row lanes; partial[rb][n]
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ a, long long lda, float* __restrict__ partial, int T,
                                                     int N, int rgrp) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (n < N) {
    for (int t = blockIdx.y * 4 + ty; t < T; t += gridDim.y * 4) {
      const long long pr = rgrp > 0 ? (long long)t + t / rgrp + 1 : (long long)t;
      s += a[pr * lda + n];
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) partial[(long long)blockIdx.y * N + n] = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
}

// The same with 16-byte loads (N, lda multiples of 4, 16-byte aligned base): block 256 = 64 column quads x 4 row lanes, a thread's rows
// requested eight at a time before the first is added (the scalar loop above had ONE 4-byte load in flight per thread: the positional
// embedding's gradient -- 512 frames x 12 800 columns, 26 MB -- took 36 us, 0.7 TB/s).  Rows are added in the same order as there.
__global__ void __launch_bounds__(256) colsum4_kernel(const float* __restrict__ a, long long lda, float* __restrict__ partial, int T,
                                                      int N, int rgrp) {
  __shared__ float4 red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + tx) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < N) {
    const int step = gridDim.y * 4;
    for (int t0 = blockIdx.y * 4 + ty; t0 < T; t0 += 8 * step) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = t0 + j * step, tc = t < T ? t : t0;
        const long long pr = rgrp > 0 ? (long long)tc + tc / rgrp + 1 : (long long)tc;
        v[j] = *reinterpret_cast<const float4*>(a + pr * lda + n);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float k = t0 + j * step < T ? 1.f : 0.f;
        s.x += v[j].x * k; s.y += v[j].y * k; s.z += v[j].z * k; s.w += v[j].w * k;
      }
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) {
    const float4 r0 = red[0][tx], r1 = red[1][tx], r2 = red[2][tx], r3 = red[3][tx];
    *reinterpret_cast<float4*>(partial + (long long)blockIdx.y * N + n) =
        make_float4(((r0.x + r1.x) + r2.x) + r3.x, ((r0.y + r1.y) + r2.y) + r3.y, ((r0.z + r1.z) + r2.z) + r3.z, ((r0.w + r1.w) + r2.w) + r3.w);
  }
}

}  // namespace

int layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int T, int D,
                  float eps, int rs, hipStream_t stream) {
  DGVIT_CHECK_ARG(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  DGVIT_CHECK_ARG(T > 0 && D > 0 && D % 4 == 0, "layernorm_fwd: D=%d must be a positive multiple of 4", D);
  DGVIT_CHECK_ARG(rs >= 1, "layernorm_fwd: bad row step");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, T, D, eps, rs);
  DGVIT_CHECK_LAUNCH("layernorm_fwd");
  return DGVIT_OK;
}

// workgroups (= partial rows of the dgamma / dbeta reduction) of the LayerNorm backward; also the scratch bound.  The row loop is
// latency-bound, so the bf16 configuration (768-wide rows, 50k of them) runs 8 workgroups per CU; the fp32 path keeps 2.
int layernorm_bwd_blocks(int T) { return T < 2048 ? (T + 3) / 4 : (T < 16384 ? 512 : 2048); }
static int layernorm_bwd_blocks_f32(int T) { return T < 2048 ? (T + 3) / 4 : 512; }

// partial is [nb][2][D]: one pass reduces both halves (stride 2*D), first D sums -> dgamma, next D -> dbeta;
// a null dgamma / dbeta (frozen parameter) is skipped
// grp != null: the reduction is queued there (the caller flushes several reductions as one launch) instead of launched here
static int ln_param_grads(const float* partial, float* dgamma, float* dbeta, int D, int nb, hipStream_t stream, ReduceGroup* grp) {
  ReduceGroup local;
  if (!grp) reduce_group_init(local);
  ReduceGroup& g = grp ? *grp : local;
  int rc = DGVIT_OK;
  if (dgamma && dbeta) rc = reduce_group_add(g, partial, dgamma, D, dbeta, 2ll * D, nb, 2ll * D, stream);
  else if (dgamma) rc = reduce_group_add(g, partial, dgamma, D, nullptr, D, nb, 2ll * D, stream);
  else if (dbeta) rc = reduce_group_add(g, partial + D, dbeta, D, nullptr, D, nb, 2ll * D, stream);
  if (rc || grp) return rc;
  return reduce_group_flush(local, stream);
}

// partial must hold layernorm_bwd_blocks(T) * 2 * D floats; dgamma/dbeta are written (not accumulated)
int layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                  float* dx, float* dgamma, float* dbeta, float* partial, int T, int D, int rs, hipStream_t stream, ReduceGroup* grp) {
  DGVIT_CHECK_ARG(dy && x && mean && rstd && gamma && dx && partial, "layernorm_bwd: null pointer");
  DGVIT_CHECK_ARG(T > 0 && D > 0 && D % 4 == 0 && D <= 1024, "layernorm_bwd: D=%d must be a multiple of 4, <= 1024", D);
  const int nb = layernorm_bwd_blocks_f32(T);
  const int nch = (D + 255) / 256;
#define LNB(NCH)                                                                                                              \
  hipLaunchKernelGGL((layernorm_bwd_kernel<NCH, float>), dim3(nb), dim3(256), 0, stream, dy, x, mean, rstd, gamma, dres, dx, \
                     (bf16_t*)nullptr, partial, T, D, rs)
  if (nch == 1) LNB(1);
  else if (nch == 2) LNB(2);
  else if (nch == 3) LNB(3);
  else LNB(4);
#undef LNB
  DGVIT_CHECK_LAUNCH("layernorm_bwd");
  return ln_param_grads(partial, dgamma, dbeta, D, nb, stream, grp);
}

// bf16 configuration: dy bf16; dx fp32 (+ optional bf16 copy dxb)
int layernorm_bwd_bf16(const bf16_t* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                       float* dx, bf16_t* dxb, float* dgamma, float* dbeta, float* partial, int T, int D, int rs, hipStream_t stream) {
  DGVIT_CHECK_ARG(dy && x && mean && rstd && gamma && dx && partial, "layernorm_bwd_bf16: null pointer");
  DGVIT_CHECK_ARG(T > 0 && D > 0 && D % 4 == 0 && D <= 1024, "layernorm_bwd_bf16: D=%d must be a multiple of 4, <= 1024", D);
  const int nb = layernorm_bwd_blocks(T);
  const int nch = (D + 255) / 256;
  const int slot = profile_begin(PROF_OTHER, 0.0, stream);
#define LNB(NCH)                                                                                                              \
  hipLaunchKernelGGL((layernorm_bwd_kernel<NCH, bf16_t>), dim3(nb), dim3(256), 0, stream, dy, x, mean, rstd, gamma, dres, dx, dxb, \
                     partial, T, D, rs)
  if (nch == 1) LNB(1);
  else if (nch == 2) LNB(2);
  else if (nch == 3) LNB(3);
  else LNB(4);
#undef LNB
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("layernorm_bwd_bf16");
  return ln_param_grads(partial, dgamma, dbeta, D, nb, stream, nullptr);
}

int rmsnorm_fwd(const float* x, long long ldx, const float* g, float* y, int B, int D, hipStream_t stream) {
  DGVIT_CHECK_ARG(x && g && y && B > 0 && D > 0, "rmsnorm_fwd: bad arguments");
  hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, x, ldx, g, y, B, D);
  DGVIT_CHECK_LAUNCH("rmsnorm_fwd");
  return DGVIT_OK;
}

int rmsnorm_bwd_blocks(int B) { return B < 256 ? (B + 3) / 4 : 64; }

int rmsnorm_bwd(const float* dy, const float* x, long long ldx, const float* g, float* dx, long long lddx, float* dg,
                float* partial, int B, int D, hipStream_t stream) {
  DGVIT_CHECK_ARG(dy && x && g && dx && partial && B > 0 && D > 0 && D <= 4096, "rmsnorm_bwd: bad arguments");
  const int nb = rmsnorm_bwd_blocks(B);
  hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(nb), dim3(256), 4 * D * sizeof(float), stream, dy, x, ldx, g, dx, lddx, partial, B, D);
  DGVIT_CHECK_LAUNCH("rmsnorm_bwd");
  return dg ? reduce_slabs(partial, dg, D, nb, D, stream) : DGVIT_OK;   // dg == null: frozen gain
}

int colsum_blocks(int T) { return T < 64 ? 1 : (T < 16384 ? 16 : 64); }

// out[n] = sum_t a[row(t)][n]; partial must hold colsum_blocks(T) * N floats
int colsum(const float* a, long long lda, float* out, float* partial, int T, int N, int rgrp, hipStream_t stream) {
  DGVIT_CHECK_ARG(a && out && partial && T > 0 && N > 0, "colsum: bad arguments");
  const int rb = colsum_blocks(T);
  if (N % 4 == 0 && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(partial) & 15) == 0)
    hipLaunchKernelGGL(colsum4_kernel, dim3((N + 255) / 256, rb), dim3(256), 0, stream, a, lda, partial, T, N, rgrp);
  else
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, rb), dim3(256), 0, stream, a, lda, partial, T, N, rgrp);
  DGVIT_CHECK_LAUNCH("colsum");
  return reduce_slabs(partial, out, N, rb, N, stream);
}
