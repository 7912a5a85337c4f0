// Small-batch forward of the GoT encoder blocks (GoalFormer.py:101-105 with :31-82): TWO launches per transformer block
// instead of seven, for the regime the reference's SAC loop lives in -- SAC.choose_action on one frame (DRL.py:170-185) and
// the no-grad passes of learn() at the shipped batch of 32 (config.yaml:11).  At T = B * N = 65 ... 2080 token rows every GEMM
// launch of the large-batch schedule is one or two workgroups' worth of latency; here the work of a block is cut by HEAD and
// by HIDDEN-COLUMN CHUNK so that B * H, resp. B * M / 128, workgroups run side by side, and each keeps its frame's tokens in LDS:
//
//   frame_attn_kernel  (one workgroup per frame and head h)
//       x      = base + bias + sum_c part_in[c]            the previous block's feed-forward partials, summed in chunk order
//       ln     = LayerNorm1(x)
//       q,k,v  = ln Wq_h^T, ln Wk_h^T, ln Wv_h^T            MFMA, weights straight from L2
//       ao_h   = softmax(q k^T / sqrt(dh)) v                scores transposed (key rows in registers), whole softmax in registers
//       part_a[h] = ao_h Wout[:, h*dh : (h+1)*dh]^T          this head's share of to_out (summed by the next kernel)
//   frame_mlp_kernel   (one workgroup per frame and chunk c of 128 hidden units)
//       xmid   = x + b_out + sum_h part_a[h]                (chunk 0 also stores it: the residual base of the next block)
//       a      = gelu(LayerNorm2(xmid) W1_c^T + b1_c)       stays in LDS
//       part_b[c] = a W2[:, c*128 : (c+1)*128]^T
//   frame_final_kernel (one wave per frame):  feat = RMSNorm(token 0 of xmid + b2 + sum_c part_b[c])
//
// Sums over heads / chunks are taken by the CONSUMER in a fixed order (deterministic, no atomics, no in-launch hand-off: the
// kernel boundary is the synchronisation).  fp32 throughout, v_mfma_f32_32x32x2_f32.
#include "common.h"
#include "kernels.h"
#include "small_mma.h"

namespace {

constexpr int FMC = 128;   // hidden columns per frame_mlp workgroup
#define DGVIT_LOG2E_F 1.4426950408889634f

struct FrameArgs {
  int B, N, NP, D, H, dh, I, M, C;
  // ---- input assembly: x = base + bias + sum_p parts[p]
  const float* base;        // (B, N, D)
  const float* bias;        // (D) or null
  const float* parts;       // (B, nparts, N, D) or null
  int nparts;
  float* x_out;             // (B, N, D): the assembled rows, written by the first workgroup of each frame (null: not needed)
  // ---- parameters of this block
  const float* lnw; const float* lnb;
  const float* wqkv; const float* wout;   // frame_attn
  const float* w1; const float* b1; const float* w2;   // frame_mlp
  float* part_out;          // (B, H | C, N, D)
  float scale;
};

// x rows of frame b -> LDS [NP][D + 4] (+ optional copy to x_out); rows >= N are zero
__device__ __forceinline__ void assemble_rows(const FrameArgs& a, float* xs, int b, bool store, int tid) {
  const int SD = a.D + 4, D4 = a.D / 4;
  const long long fbase = (long long)b * a.N * a.D;
  for (int f = tid; f < a.NP * D4; f += 256) {
    const int row = f / D4, c = (f % D4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < a.N) {
      v = *reinterpret_cast<const float4*>(a.base + fbase + (long long)row * a.D + c);
      if (a.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(a.bias + c);
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      }
      const float* pp = a.parts + ((long long)b * a.nparts * a.N + row) * a.D + c;
      const long long pstride = (long long)a.N * a.D;
      for (int p0 = 0; p0 < a.nparts; p0 += 8) {      // 8 partial loads in flight, added in part order
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          t[u] = p0 + u < a.nparts ? *reinterpret_cast<const float4*>(pp + (p0 + u) * pstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w;
        }
      }
      if (store) *reinterpret_cast<float4*>(a.x_out + fbase + (long long)row * a.D + c) = v;
    }
    *reinterpret_cast<float4*>(xs + row * SD + c) = v;
  }
}

// LayerNorm (eps 1e-5, affine) of rows [0, N) of the LDS image, in place; one wave per row
__device__ __forceinline__ void layernorm_rows(const FrameArgs& a, float* xs, int tid) {
  const int SD = a.D + 4, lane = tid & 63, wave = tid >> 6;
  const int c = lane * 4;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f), bt = g;
  if (c < a.D) {
    g = *reinterpret_cast<const float4*>(a.lnw + c);
    bt = *reinterpret_cast<const float4*>(a.lnb + c);
  }
  const float invD = 1.f / (float)a.D;
  for (int row = wave; row < a.N; row += 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < a.D) v = *reinterpret_cast<const float4*>(xs + row * SD + c);
    const float mu = wave_sum((v.x + v.y) + (v.z + v.w)) * invD;
    const float d0 = v.x - mu, d1 = v.y - mu, d2 = v.z - mu, d3 = v.w - mu;
    const float q = c < a.D ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
    const float rs = rsqrtf(wave_sum(q) * invD + 1e-5f);
    if (c < a.D)
      *reinterpret_cast<float4*>(xs + row * SD + c) = make_float4(d0 * rs * g.x + bt.x, d1 * rs * g.y + bt.y, d2 * rs * g.z + bt.z, d3 * rs * g.w + bt.w);
  }
}

// ------------------------------------------------------------------------------------------------ attention half of a block
template <int NKT>   // 32-key tiles (NP = 32 * NKT)
__global__ void __launch_bounds__(256) frame_attn_kernel(const FrameArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int SD = a.D + 4, SH = a.dh + 4, NP = 32 * NKT;
  float* xs = smem;
  float* qs = xs + NP * SD;     // q (pre-scaled), later this head's attention output
  float* ks = qs + NP * SH;
  float* vs = ks + NP * SH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / a.H, hd = blockIdx.x % a.H;
  assemble_rows(a, xs, b, hd == 0 && a.x_out, tid);
  __syncthreads();
  layernorm_rows(a, xs, tid);
  __syncthreads();
  // q, k, v of this head: 3 matrices x (dh / 32) column tiles x NKT row tiles of 32 x 32, K = D
  const int ct_n = a.dh / 32, nblk = 3 * ct_n * NKT;
  const float qscale = a.scale * DGVIT_LOG2E_F;
  for (int blk = wave; blk < nblk; blk += 4) {
    const int rt = blk % NKT, ct = (blk / NKT) % ct_n, mat = blk / (NKT * ct_n);
    f32x16 acc;
    zero16(acc);
    const int jn = ct * 32 + li;
    mm_rows_x_wrows<true>(acc, xs + rt * 32 * SD, SD, a.wqkv + ((long long)mat * a.I + hd * a.dh) * a.D, a.D, jn, a.dh, a.D, a.D, li, h);
    float* dst = mat == 0 ? qs : (mat == 1 ? ks : vs);
    const float mul = mat == 0 ? qscale : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(rt * 32 + arow(r, h)) * SH + jn] = acc[r] * mul;
  }
  __syncthreads();
  // attention: one wave per 32-query tile; S^T[key][query] so that the softmax over keys is register-local
  for (int qt = wave; qt < NKT; qt += 4) {
    if (qt * 32 >= a.N) break;
    const float* qrow = qs + (qt * 32 + li) * SH;
    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) zero16(s[kt]);
    for (int g = 0; g < a.dh / 8; ++g) {
      const float4 qf = *reinterpret_cast<const float4*>(qrow + 8 * g + 4 * h);
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const float4 kf = *reinterpret_cast<const float4*>(ks + (kt * 32 + li) * SH + 8 * g + 4 * h);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf.x, s[kt], 0, 0, 0);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf.y, s[kt], 0, 0, 0);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf.z, s[kt], 0, 0, 0);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf.w, s[kt], 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = kt * 32 + arow(r, h) < a.N ? s[kt][r] : -INFINITY;
        s[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[kt][r] - mx);
        s[kt][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    // O^T[d][query] = sum_key V[key][d] P^T[key][query]: the probability registers are the B operand as they stand
    for (int dt = 0; dt < ct_n; ++dt) {
      f32x16 o;
      zero16(o);
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          o = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[(kt * 32 + arow(r, h)) * SH + dt * 32 + li], s[kt][r], o, 0, 0, 0);
      // this wave has its q fragments in flight no more: its rows of `qs` now take the attention output
      float* orow = qs + (qt * 32 + li) * SH + dt * 32;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<float4*>(orow + 8 * c + 4 * h) = make_float4(o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv);
    }
  }
  __syncthreads();
  // this head's share of to_out: part[n][j] = sum_d ao[n][d] Wout[j][hd * dh + d]
  float* part = a.part_out + ((long long)b * a.H + hd) * a.N * a.D;
  const int jt_n = a.D / 32;
  for (int blk = wave; blk < NKT * jt_n; blk += 4) {
    const int rt = blk % NKT, jt = blk / NKT;
    if (rt * 32 >= a.N) continue;
    f32x16 acc;
    zero16(acc);
    const int jn = jt * 32 + li;
    mm_rows_x_wrows<true>(acc, qs + rt * 32 * SH, SH, a.wout + hd * a.dh, a.I, jn, a.D, a.dh, a.dh, li, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rt * 32 + arow(r, h);
      if (row < a.N) part[(long long)row * a.D + jn] = acc[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------ feed-forward half of a block
template <int NKT>
__global__ void __launch_bounds__(256) frame_mlp_kernel(const FrameArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int SD = a.D + 4, SA = FMC + 4, NP = 32 * NKT;
  float* xs = smem;
  float* as = xs + NP * SD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / a.C, c = blockIdx.x % a.C;
  assemble_rows(a, xs, b, c == 0 && a.x_out, tid);
  __syncthreads();
  layernorm_rows(a, xs, tid);
  __syncthreads();
  // hidden chunk: a = gelu(ln W1_c^T + b1_c), 32 x 32 blocks: NKT row tiles x 4 column tiles, K = D
  for (int blk = wave; blk < NKT * (FMC / 32); blk += 4) {
    const int rt = blk % NKT, ct = blk / NKT;
    f32x16 acc;
    zero16(acc);
    const int jn = ct * 32 + li;
    const float bias = a.b1[c * FMC + jn];
    mm_rows_x_wrows<true>(acc, xs + rt * 32 * SD, SD, a.w1 + (long long)c * FMC * a.D, a.D, jn, FMC, a.D, a.D, li, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) as[(rt * 32 + arow(r, h)) * SA + jn] = gelu_erf(acc[r] + bias);
  }
  __syncthreads();
  // part[n][j] = sum_m a[n][m] W2[j][c * 128 + m]
  float* part = a.part_out + ((long long)b * a.C + c) * a.N * a.D;
  const int jt_n = a.D / 32;
  for (int blk = wave; blk < NKT * jt_n; blk += 4) {
    const int rt = blk % NKT, jt = blk / NKT;
    if (rt * 32 >= a.N) continue;
    f32x16 acc;
    zero16(acc);
    const int jn = jt * 32 + li;
    mm_rows_x_wrows<true>(acc, as + rt * 32 * SA, SA, a.w2 + c * FMC, a.M, jn, a.D, FMC, FMC, li, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rt * 32 + arow(r, h);
      if (row < a.N) part[(long long)row * a.D + jn] = acc[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------ pooled token + RMSNorm
// feat[b] = F.normalize(v) * sqrt(D) * g with v = token 0 of (base + bias + sum_c parts[c])   (GoalFormer.py:167-170); one wave per frame
__global__ void __launch_bounds__(256) frame_final_kernel(const float* __restrict__ base, const float* __restrict__ bias,
                                                          const float* __restrict__ parts, int nparts, const float* __restrict__ g,
                                                          float* __restrict__ feat, int B, int N, int D) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float v[4] = {0.f, 0.f, 0.f, 0.f};   // D <= 256: columns lane, lane + 64, ...
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < D) {
      float s = base[(long long)b * N * D + c] + bias[c];
      for (int p = 0; p < nparts; ++p) s += parts[((long long)b * nparts + p) * N * D + c];
      v[i] = s;
      q += s * s;
    }
  }
  const float n = fmaxf(sqrtf(wave_sum(q)), 1e-12f), sc = sqrtf((float)D);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < D) feat[(long long)b * D + c] = v[i] / n * sc * g[c];
  }
}

size_t attn_lds(int NP, int D, int dh) { return sizeof(float) * ((size_t)NP * (D + 4) + 3 * (size_t)NP * (dh + 4)); }
size_t mlp_lds(int NP, int D) { return sizeof(float) * ((size_t)NP * (D + 4) + (size_t)NP * (FMC + 4)); }

template <int NKT>
int launch_pair(const FrameArgs& aa, const FrameArgs& ab, hipStream_t st) {
  const size_t la = attn_lds(aa.NP, aa.D, aa.dh), lb = mlp_lds(ab.NP, ab.D);
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(frame_attn_kernel<NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlp_kernel<NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "frame kernels: hipFuncSetAttribute failed");
    once.mark(bit);
  }
  const int slot = profile_begin(PROF_OTHER, 0.0, st);
  hipLaunchKernelGGL(frame_attn_kernel<NKT>, dim3(aa.B * aa.H), dim3(256), la, st, aa);
  hipLaunchKernelGGL(frame_mlp_kernel<NKT>, dim3(ab.B * ab.C), dim3(256), lb, st, ab);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("frame kernels");
  return DGVIT_OK;
}

}  // namespace

// Can the two-launches-per-block path run this shape?  (token count, widths, LDS footprint)
bool frame_path_supports(int B, int N, int D, int H, int dh, int M) {
  const int NP = (N + 31) / 32 * 32;
  if (B <= 0 || N <= 0 || NP > 128 || D <= 0 || D > 256 || D % 32 != 0 || (dh != 32 && dh != 64) || M % FMC != 0 || H <= 0) return false;
  return attn_lds(NP, D, dh) <= 160 * 1024 && mlp_lds(NP, D) <= 160 * 1024;
}

// floats of scratch: x (T D) | xmid (T D) | part_a (H T D) | part_b (C T D)
long long frame_path_scratch_floats(int B, int N, int D, int H, int M) {
  const long long TD = (long long)B * N * D;
  return TD * (2 + H + M / FMC);
}

// x0 (B, N, D): the assembled, dropped-out token rows; params: the table of dgvit_got_forward; writes feat (B, D)
int frame_path_forward(const float* x0, const float* const* params, int L, float* scratch, float* feat, int B, int N, int D, int H, int dh,
                       int M, hipStream_t st) {
  DGVIT_CHECK_ARG(frame_path_supports(B, N, D, H, dh, M), "frame path: unsupported shape");
  const long long TD = (long long)B * N * D;
  const int C = M / FMC, NP = (N + 31) / 32 * 32;
  float* X = scratch;
  float* XM = X + TD;
  float* PA = XM + TD;
  float* PB = PA + (long long)H * TD;
  enum { L_LN1W = 0, L_LN1B, L_QKV, L_OUTW, L_OUTB, L_LN2W, L_LN2B, L_FC1W, L_FC1B, L_FC2W, L_FC2B, PER = 11, P_RMS = 3, P_L0 = 4 };
  for (int i = 0; i < L; ++i) {
    const float* const* lp = params + P_L0 + PER * i;
    FrameArgs aa = {};
    aa.B = B; aa.N = N; aa.NP = NP; aa.D = D; aa.H = H; aa.dh = dh; aa.I = H * dh; aa.M = M; aa.C = C;
    aa.scale = 1.0f / sqrtf((float)dh);
    if (i == 0) { aa.base = x0; aa.bias = nullptr; aa.parts = nullptr; aa.nparts = 0; }
    else { aa.base = XM; aa.bias = params[P_L0 + PER * (i - 1) + L_FC2B]; aa.parts = PB; aa.nparts = C; }
    aa.x_out = X;
    aa.lnw = lp[L_LN1W]; aa.lnb = lp[L_LN1B]; aa.wqkv = lp[L_QKV]; aa.wout = lp[L_OUTW];
    aa.part_out = PA;
    FrameArgs ab = aa;
    ab.base = X; ab.bias = lp[L_OUTB]; ab.parts = PA; ab.nparts = H;
    ab.x_out = XM;
    ab.lnw = lp[L_LN2W]; ab.lnb = lp[L_LN2B]; ab.w1 = lp[L_FC1W]; ab.b1 = lp[L_FC1B]; ab.w2 = lp[L_FC2W];
    ab.part_out = PB;
    int rc;
    switch (NP / 32) {
      case 1: rc = launch_pair<1>(aa, ab, st); break;
      case 2: rc = launch_pair<2>(aa, ab, st); break;
      case 3: rc = launch_pair<3>(aa, ab, st); break;
      default: rc = launch_pair<4>(aa, ab, st); break;
    }
    if (rc) return rc;
  }
  hipLaunchKernelGGL(frame_final_kernel, dim3((B + 3) / 4), dim3(256), 0, st, XM, params[P_L0 + PER * (L - 1) + L_FC2B], PB, C, params[P_RMS], feat, B, N, D);
  DGVIT_CHECK_LAUNCH("frame_final_kernel");
  return DGVIT_OK;
}
