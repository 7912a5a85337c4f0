// Fused multi-head self-attention for short sequences (N <= 224 tokens), fp32 on v_mfma_f32_32x32x2_f32.
// Restates Attention.forward, GoalFormer.py:73-81: per (frame, head)  softmax(q k^T * dh^-1/2) v, reading q/k/v
// straight out of the (B, N, 3*I) to_qkv output ([q heads | k heads | v heads], 64 columns per head) and writing
// the merged-head (B, N, I) layout -- the reference's two einops rearrange copies never materialise.
//
// One workgroup per (frame, head); the whole K and V of the head sit in LDS; one wave per 32-query tile.
// Scores are computed TRANSPOSED (S^T[key][query] = K Q^T) so that a lane owns one query column and the 32x32
// accumulator registers hold keys: the row softmax is register-local plus one cross-half exchange, and the
// accumulator registers are directly the B operand of the next product (O^T = V^T P^T; dQ^T = K^T dS^T), with no
// LDS round trip (step r contracts keys (r&3) + 8*(r>>2) + 4*half).
//
// Backward recomputes the probabilities (nothing but q/k/v/o is saved):
//   phase 1 (wave = query tile, K/V in LDS):  P^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta) * scale, dQ^T = K^T dS^T
//   phase 2 (wave = key tile, Q/dO in LDS):   P = exp(S*scale - lse), dP = dO V^T, dV^T = dO^T P, dK^T = Q^T dS
// with delta[q] = sum_d dO[q][d] O[q][d] and lse[q] handed from phase 1 to phase 2 through LDS.
#include "common.h"

namespace {

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// softmax runs in base 2: the query is pre-scaled by scale*log2(e), so exp(x - max) = exp2(s' - max') on v_exp_f32
#define DGVIT_LOG2E 1.4426950408889634f
#define DGVIT_LN2 0.6931471805599453f

// Stage rows [0, nrows) of two 64-wide (DH-wide) per-head column blocks into LDS images [NP][SK], zero padding
// rows.  Fully unrolled for the compile-time thread count: every thread first issues ALL its global loads
// (2 * NP*DH/4/NTHR float4 in flight), then writes LDS -- a runtime-trip-count loop here serialises one memory
// round trip per float4.  Out-of-range rows read row 0 and are zeroed by a select (no divergent branches).
template <int DH, int SK, int NP, int NTHR>
__device__ __forceinline__ void stage_pair(float* dstA, const float* srcA, long long ldA, float* dstB, const float* srcB,
                                           long long ldB, int nrows, int tid) {
  constexpr int C4 = DH / 4;
  constexpr int ITER = (NP * C4 + NTHR - 1) / NTHR;
  constexpr int CH = 4;   // 2*CH float4 (32 VGPRs) in flight per thread: enough to cover the latency, no spills
#pragma unroll
  for (int i0 = 0; i0 < ITER; i0 += CH) {
    float4 va[CH], vb[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int f = tid + (i0 + j) * NTHR;
      const int row = f / C4, c = (f % C4) * 4;
      const int rr = (i0 + j < ITER && row < nrows) ? row : 0;
      va[j] = *reinterpret_cast<const float4*>(srcA + rr * ldA + c);
      vb[j] = *reinterpret_cast<const float4*>(srcB + rr * ldB + c);
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int f = tid + (i0 + j) * NTHR;
      const int row = f / C4, c = (f % C4) * 4;
      if (i0 + j < ITER && (NP * C4 % NTHR == 0 || row < NP)) {
        const float k = row < nrows ? 1.f : 0.f;   // (a float4 ?: would be lowered through scratch memory)
        *reinterpret_cast<float4*>(dstA + row * SK + c) = make_float4(va[j].x * k, va[j].y * k, va[j].z * k, va[j].w * k);
        *reinterpret_cast<float4*>(dstB + row * SK + c) = make_float4(vb[j].x * k, vb[j].y * k, vb[j].z * k, vb[j].w * k);
      }
    }
  }
}

// B-operand style fragments of one row (lane owns a row): elements [8g + 4h .. +3], g = 0..DH/8.
// `rowptr` must point at a readable row (callers clamp the row index); invalid rows are zeroed by a select.
template <int DH>
__device__ __forceinline__ void row_frags(float4 (&f)[DH / 8], const float* rowptr, bool valid, int h, float mul) {
  const float m = valid ? mul : 0.f;
#pragma unroll
  for (int g = 0; g < DH / 8; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(rowptr + 8 * g + 4 * h);
    f[g] = make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
  }
}

// acc += rowsA(LDS image, rows base+li) . fragsB   over the DH-deep contraction
template <int DH, int SK>
__device__ __forceinline__ void mfma_rows_x_frags(f32x16& acc, const float* img, int row, int h, const float4 (&fb)[DH / 8]) {
#pragma unroll
  for (int g = 0; g < DH / 8; ++g) {
    const float4 a = *reinterpret_cast<const float4*>(img + row * SK + 8 * g + 4 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, fb[g].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, fb[g].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, fb[g].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, fb[g].w, acc, 0, 0, 0);
  }
}

// transposed accumulator tile (rows = d, cols = token on the lane) -> global row `tok`, 16-byte pieces along d
template <int DH>
__device__ __forceinline__ void store_T(const f32x16 (&o)[DH / 32], float* rowptr, int h, float mul) {
#pragma unroll
  for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 v = make_float4(o[dt][4 * c] * mul, o[dt][4 * c + 1] * mul, o[dt][4 * c + 2] * mul, o[dt][4 * c + 3] * mul);
      *reinterpret_cast<float4*>(rowptr + dt * 32 + 8 * c + 4 * h) = v;
    }
}

// ------------------------------------------------------------------------------------ forward
template <int DH, int NKT>
__global__ void __launch_bounds__(64 * (NKT < 4 ? NKT : 4), NKT <= 2 ? 2 : 1) attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int H,
                                                       float scale, int nq) {
  constexpr int SK = DH + 4, NP = NKT * 32, DT = DH / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = smem + NP * SK;
  constexpr int NW = NKT < 4 ? NKT : 4, NTHR = 64 * NW;   // launch configuration (see launch_fwd)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = NW;
  const int li = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int I = H * DH;
  const long long ld = 3ll * I;
  const float* base = qkv + (long long)b * N * ld + hd * DH;

  const int nqt = (nq + 31) / 32;   // only queries < nq are needed (nq = 1: the last block keeps token 0 only)
  // the per-lane Q fragments are requested BEFORE the K/V staging so both global round trips overlap
  float4 qf[DH / 8];
  {
    const int q0 = wave * 32 + li;
    row_frags<DH>(qf, base + (q0 < nq ? q0 : 0) * ld, q0 < nq, h, scale * DGVIT_LOG2E);
  }
  stage_pair<DH, SK, NP, NTHR>(Ks, base + I, ld, Vs, base + 2 * I, ld, N, tid);
  __syncthreads();

  for (int qt = wave; qt < nqt; qt += nw) {
    const int q = qt * 32 + li;

    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      mfma_rows_x_frags<DH, SK>(s[kt], Ks, kt * 32 + li, h, qf);
    }
    if (qt + nw < nqt)   // next tile's Q fragments travel while this tile's softmax and P.V run
    {
      const int qn = (qt + nw) * 32 + li;
      row_frags<DH>(qf, base + (qn < nq ? qn : 0) * ld, qn < nq, h, scale * DGVIT_LOG2E);
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (kt * 32 + acc_row(r, h) < N) ? s[kt][r] : -INFINITY;
        s[kt][r] = v;
        m = fmaxf(m, v);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[kt][r] - m);
        s[kt][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 32, 64);

    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vrow = Vs + (kt * 32 + acc_row(r, h)) * SK + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[dt * 32], s[kt][r], o[dt], 0, 0, 0);
      }
    if (q < nq) store_T<DH>(o, out + ((long long)b * N + q) * I + hd * DH, h, 1.f / l);
  }
}

// ------------------------------------------------------------------------------------ backward
template <int DH, int NKT>
__global__ void __launch_bounds__(64 * (NKT < 4 ? NKT : 4), NKT <= 2 ? 2 : 1) attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ o_fwd,
                                                       const float* __restrict__ d_out, float* __restrict__ dqkv, int N, int H,
                                                       float scale, int nq) {
  constexpr int SK = DH + 4, NP = NKT * 32, DT = DH / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X = smem;                 // phase 1: K      phase 2: Q
  float* Y = smem + NP * SK;       // phase 1: V      phase 2: dO
  float* lse_s = smem + 2 * NP * SK;
  float* del_s = lse_s + NP;
  constexpr int NW = NKT < 4 ? NKT : 4, NTHR = 64 * NW;   // launch configuration (see launch_bwd)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = NW;
  const int li = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int I = H * DH;
  const long long ld = 3ll * I;
  const float* base = qkv + (long long)b * N * ld + hd * DH;
  const float* obase = o_fwd + (long long)b * N * I + hd * DH;
  const float* dobase = d_out + (long long)b * N * I + hd * DH;
  float* gbase = dqkv + (long long)b * N * ld + hd * DH;

  const int nqt = (nq + 31) / 32;
  float4 qf[DH / 8], dof[DH / 8], of[DH / 8];
  {  // first query tile's per-lane fragments are requested before the K/V staging (overlapping round trips)
    const int q0 = wave * 32 + li;
    const bool v0 = q0 < nq;
    const int qc = v0 ? q0 : 0;
    row_frags<DH>(qf, base + qc * ld, v0, h, scale * DGVIT_LOG2E);
    row_frags<DH>(dof, dobase + (long long)qc * I, v0, h, 1.f);
    row_frags<DH>(of, obase + (long long)qc * I, v0, h, 1.f);
  }
  stage_pair<DH, SK, NP, NTHR>(X, base + I, ld, Y, base + 2 * I, ld, N, tid);
  __syncthreads();

  // ---- phase 1: one query tile per wave -> dQ, lse, delta
  for (int qt = wave; qt < nqt; qt += nw) {
    const int q = qt * 32 + li;
    const bool qv = q < nq;
    if (qt != wave) {
      const int qc = qv ? q : 0;
      row_frags<DH>(qf, base + qc * ld, qv, h, scale * DGVIT_LOG2E);
      row_frags<DH>(dof, dobase + (long long)qc * I, qv, h, 1.f);
      row_frags<DH>(of, obase + (long long)qc * I, qv, h, 1.f);
    }
    float delta = 0.f;
#pragma unroll
    for (int g = 0; g < DH / 8; ++g)
      delta += (dof[g].x * of[g].x + dof[g].y * of[g].y) + (dof[g].z * of[g].z + dof[g].w * of[g].w);
    delta += __shfl_xor(delta, 32, 64);

    f32x16 p[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) p[kt][r] = 0.f;
      mfma_rows_x_frags<DH, SK>(p[kt], X, kt * 32 + li, h, qf);
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (kt * 32 + acc_row(r, h) < N) ? p[kt][r] : -INFINITY;
        p[kt][r] = v;
        m = fmaxf(m, v);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(p[kt][r] - m);
        p[kt][r] = e;
        l += e;
      }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    if (h == 0) {
      lse_s[q] = m + __builtin_amdgcn_logf(l);   // base-2 log-sum-exp of the base-2 scores
      del_s[q] = delta;
    }

    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x16 dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
      mfma_rows_x_frags<DH, SK>(dp, Y, kt * 32 + li, h, dof);  // dP^T[key][q] = sum_d V[key][d] dO[q][d]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float ds = p[kt][r] * inv * (dp[r] - delta) * scale;
        const float* krow = X + (kt * 32 + acc_row(r, h)) * SK + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[dt * 32], ds, dq[dt], 0, 0, 0);
      }
    }
    if (qv) store_T<DH>(dq, gbase + q * ld, h, 1.f);
  }
  // ---- phase 2: Q and dO into LDS, one key tile per wave -> dK, dV
  float4 kf[DH / 8], vf[DH / 8];
  {  // first key tile's fragments are requested before the phase barrier and the restaging
    const int k0 = wave * 32 + li;
    const bool v0 = k0 < N;
    const int kc = v0 ? k0 : 0;
    row_frags<DH>(kf, base + I + kc * ld, v0, h, 1.f);
    row_frags<DH>(vf, base + 2 * I + kc * ld, v0, h, 1.f);
  }
  __syncthreads();
  stage_pair<DH, SK, NP, NTHR>(X, base, ld, Y, dobase, (long long)I, nq, tid);   // rows >= nq zero-filled: no gradient
  __syncthreads();
  for (int kt = wave; kt < NKT; kt += nw) {
    const int key = kt * 32 + li;
    const bool kv = key < N;
    if (kt != wave) {
      const int kc = kv ? key : 0;
      row_frags<DH>(kf, base + I + kc * ld, kv, h, 1.f);
      row_frags<DH>(vf, base + 2 * I + kc * ld, kv, h, 1.f);
    }
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dk[dt][r] = 0.f;
        dv[dt][r] = 0.f;
      }
#pragma unroll 1
    for (int qt = 0; qt < nqt; ++qt) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = 0.f;
        dp[r] = 0.f;
      }
      mfma_rows_x_frags<DH, SK>(s, X, qt * 32 + li, h, kf);   // S[q][key]
      mfma_rows_x_frags<DH, SK>(dp, Y, qt * 32 + li, h, vf);  // dP[q][key]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qt * 32 + acc_row(r, h);
        const float pv = (kv && q < nq) ? __builtin_amdgcn_exp2f(s[r] * (scale * DGVIT_LOG2E) - lse_s[q]) : 0.f;
        const float ds = pv * (dp[r] - del_s[q]) * scale;
        const float* dorow = Y + q * SK + li;
        const float* qrow = X + q * SK + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dorow[dt * 32], pv, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[dt * 32], ds, dk[dt], 0, 0, 0);
        }
      }
    }
    if (kv) {
      store_T<DH>(dk, gbase + I + key * ld, h, 1.f);
      store_T<DH>(dv, gbase + 2 * I + key * ld, h, 1.f);
    }
  }
}

template <int DH, int NKT>
int launch_fwd(const float* qkv, float* out, int B, int N, int H, float scale, int nq, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * NKT * 32 * (DH + 4) * sizeof(float);
  auto kern = attn_fwd_kernel<DH, NKT>;
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, "attention_fwd: %s", hipGetErrorString(e));
    done = true;
  }
  const int nw = NKT < 4 ? NKT : 4;
  const int slot = profile_begin(PROF_ATTN_FWD, 4.0 * B * H * (double)nq * N * DH, stream);
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(64 * nw), lds, stream, qkv, out, N, H, scale, nq);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("attention_fwd");
  return DGVIT_OK;
}

template <int DH, int NKT>
int launch_bwd(const float* qkv, const float* o, const float* dout, float* dqkv, int B, int N, int H, float scale, int nq,
               hipStream_t stream) {
  constexpr size_t lds = ((size_t)2 * NKT * 32 * (DH + 4) + 2 * NKT * 32) * sizeof(float);
  auto kern = attn_bwd_kernel<DH, NKT>;
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, "attention_bwd: %s", hipGetErrorString(e));
    done = true;
  }
  const int nw = NKT < 4 ? NKT : 4;
  const int slot = profile_begin(PROF_ATTN_BWD, 8.0 * B * H * (double)nq * N * DH, stream);
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(64 * nw), lds, stream, qkv, o, dout, dqkv, N, H, scale, nq);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("attention_bwd");
  return DGVIT_OK;
}

}  // namespace

#define ATTN_DISPATCH(FN, ...)                                                        \
  switch (dh * 8 + nkt) {                                                             \
    case 64 * 8 + 1: return FN<64, 1>(__VA_ARGS__);                                   \
    case 64 * 8 + 2: return FN<64, 2>(__VA_ARGS__);                                   \
    case 64 * 8 + 3: return FN<64, 3>(__VA_ARGS__);                                   \
    case 64 * 8 + 4: return FN<64, 4>(__VA_ARGS__);                                   \
    case 64 * 8 + 5: return FN<64, 5>(__VA_ARGS__);                                   \
    case 64 * 8 + 6: return FN<64, 6>(__VA_ARGS__);                                   \
    case 64 * 8 + 7: return FN<64, 7>(__VA_ARGS__);                                   \
    case 32 * 8 + 1: return FN<32, 1>(__VA_ARGS__);                                   \
    case 32 * 8 + 2: return FN<32, 2>(__VA_ARGS__);                                   \
    default: break;                                                                   \
  }

// qkv (B, N, 3*H*dh) -> out (B, N, H*dh)
// nq = number of leading query tokens whose output is needed (N normally)
int attention_fwd(const float* qkv, float* out, int B, int N, int H, int dh, int nq, hipStream_t stream) {
  DGVIT_CHECK_ARG(qkv && out && B > 0 && N > 0 && H > 0, "attention_fwd: bad arguments");
  DGVIT_CHECK_ARG((long long)B * H < (1ll << 31), "attention_fwd: B*H too large");
  const int nkt = (N + 31) / 32;
  const float scale = 1.0f / sqrtf((float)dh);
  DGVIT_CHECK_ARG(nq >= 1 && nq <= N, "attention_fwd: bad query count");
  ATTN_DISPATCH(launch_fwd, qkv, out, B, N, H, scale, nq, stream)
  return dgvit_set_error(DGVIT_ERR_ARG, "attention_fwd: unsupported dim_head=%d / tokens=%d (dim_head 64 with N<=224, or 32 with N<=64)", dh, N);
}

// dqkv (B, N, 3*H*dh) is fully written for rows < N
// with nq < N only rows < nq of `o`/`dout` are read and only rows < nq of dq are written (dk, dv: all rows)
int attention_bwd(const float* qkv, const float* o, const float* dout, float* dqkv, int B, int N, int H, int dh, int nq,
                  hipStream_t stream) {
  DGVIT_CHECK_ARG(qkv && o && dout && dqkv && B > 0 && N > 0 && H > 0, "attention_bwd: bad arguments");
  DGVIT_CHECK_ARG((long long)B * H < (1ll << 31), "attention_bwd: B*H too large");
  const int nkt = (N + 31) / 32;
  const float scale = 1.0f / sqrtf((float)dh);
  DGVIT_CHECK_ARG(nq >= 1 && nq <= N, "attention_bwd: bad query count");
  ATTN_DISPATCH(launch_bwd, qkv, o, dout, dqkv, B, N, H, scale, nq, stream)
  return dgvit_set_error(DGVIT_ERR_ARG, "attention_bwd: unsupported dim_head=%d / tokens=%d", dh, N);
}
