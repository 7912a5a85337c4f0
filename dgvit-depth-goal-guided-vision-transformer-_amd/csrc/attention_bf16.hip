// Fused multi-head self-attention, bf16 storage / fp32 softmax and accumulation on v_mfma_f32_32x32x16_bf16
// (Attention.forward, GoalFormer.py:73-81, in the bf16 configuration).  Same plan as attention.hip:
// one workgroup per (frame, head), K and V of the head in LDS, one wave per 32-query tile, scores computed
// transposed (S^T[key][query] = K Q^T) so that a lane owns one query: the softmax is register-local plus one
// cross-half exchange and the probability tile, converted to bf16 pairwise, IS the B operand of O^T = V^T P^T
// (registers 8s..8s+7 form k-step s; element j of lane half h is key 16 s + 8 (j >> 2) + 4 h + (j & 3)).
//
// LDS images:
//   K  [NP][64] bf16, 128-byte rows, 16-byte chunks XOR-swizzled by ((key >> 1) & 7)  -> conflict-free ds_read_b128
//   Vt [64][NP + 8] bf16: V transposed at staging time (the contraction index of P V is the key, so V must be
//      k-contiguous per feature row); inside each 16-key group keys are stored at position 8*((k>>2)&1) + 4*(k>>3) + (k&3)
//      so that the eight keys a lane needs for one k-step are 16 contiguous bytes.  Row stride (2 NP + 16) bytes =
//      16 x odd: the 16 lanes of a ds_read_b128 group hit 16 distinct bank slots.
#include "bf16.h"
#include "kernels.h"

namespace {

#define DGVIT_LOG2E 1.4426950408889634f
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ int vt_pos(int key) {
  const int w = key & 15;
  return (key & ~15) | (((w >> 2) & 1) << 3) | ((w >> 3) << 2) | (w & 3);
}

template <int NW>
__global__ void __launch_bounds__(64 * NW) attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                float* __restrict__ lse, int N, int H, float scale, int nq) {
  constexpr int DH = 64, NTHR = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nkt = (N + 31) / 32, NP = nkt * 32, VS = NP + 8;   // VS: Vt row stride in elements
  unsigned char* Ks = smem;
  bf16_t* Vt = reinterpret_cast<bf16_t*>(smem + NP * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int I = H * DH;
  const long long ld = 3ll * I;
  const bf16_t* base = qkv + (long long)b * N * ld + hd * DH;
  const float sc = scale * DGVIT_LOG2E;

  // ---- stage K (swizzled rows) and V (transposed), zero padding keys >= N ------------------------------------
  for (int f = tid; f < NP * 8; f += NTHR) {
    const int row = f >> 3, pc = f & 7, c = pc ^ ((row >> 1) & 7);
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (row < N) v = *reinterpret_cast<const u32x4_t*>(base + I + row * ld + c * 8);
    *reinterpret_cast<u32x4_t*>(Ks + row * 128 + pc * 16) = v;
  }
  for (int f = tid; f < (NP / 2) * 8; f += NTHR) {
    const int kp = f >> 3, dc = f & 7, key = kp * 2;   // keys (key, key + 1) are adjacent in the permuted order
    u32x4_t v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
    if (key < N) v0 = *reinterpret_cast<const u32x4_t*>(base + 2 * I + key * ld + dc * 8);
    if (key + 1 < N) v1 = *reinterpret_cast<const u32x4_t*>(base + 2 * I + (key + 1) * ld + dc * 8);
    unsigned* dst = reinterpret_cast<unsigned*>(Vt + (dc * 8) * VS + vt_pos(key));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dst[(2 * i) * (VS / 2)] = (v0[i] & 0xFFFFu) | (v1[i] << 16);
      dst[(2 * i + 1) * (VS / 2)] = (v0[i] >> 16) | (v1[i] & 0xFFFF0000u);
    }
  }
  __syncthreads();

  const unsigned fsw = (unsigned)((li >> 1) & 7);
  const int nqt = (nq + 31) / 32;
  for (int qt = wave; qt < nqt; qt += NW) {
    const int q = qt * 32 + li;
    bf16x8 qf[4];
    {
      const bf16_t* qrow = base + (long long)(q < nq ? q : 0) * ld;
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s + 8 * h);
    }
    float m = -INFINITY, l = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o[0][r] = 0.f;
      o[1][r] = 0.f;
    }
    for (int kt = 0; kt < nkt; ++kt) {
      f32x16 s0;
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[r] = 0.f;
      const unsigned char* krow = Ks + (kt * 32 + li) * 128;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(krow + (((2 * s + h) ^ fsw) * 16));
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], s0, 0, 0, 0);
      }
      float mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + acc_row(r, h);
        const float v = key < N ? s0[r] * sc : -INFINITY;
        s0[r] = v;
        mt = fmaxf(mt, v);
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float mn = fmaxf(m, mt);                       // every tile holds at least one real key: finite
      const float alpha = __builtin_amdgcn_exp2f(m - mn);  // first tile: exp2(-inf) = 0
      float ts = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(s0[r] - mn);
        s0[r] = pr;
        ts += pr;
      }
      ts += __shfl_xor(ts, 32, 64);
      l = l * alpha + ts;
      m = mn;
      if (kt > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o[0][r] *= alpha;
          o[1][r] *= alpha;
        }
      }
      bf16x8 pf[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[s][j] = (__bf16)s0[8 * s + j];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(Vt + (dt * 32 + li) * VS + kt * 32 + 16 * s + 8 * h);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[s], o[dt], 0, 0, 0);
        }
    }
    if (q < nq) {
      const float inv = 1.f / l;
      bf16_t* orow = out + ((long long)b * N + q) * I + hd * DH;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          fx4 v = {o[dt][4 * c] * inv, o[dt][4 * c + 1] * inv, o[dt][4 * c + 2] * inv, o[dt][4 * c + 3] * inv};
          *reinterpret_cast<bf16x4*>(orow + dt * 32 + 8 * c + 4 * h) = __builtin_convertvector(v, bf16x4);
        }
      if (lse && h == 0) lse[((long long)b * H + hd) * N + q] = m + __builtin_amdgcn_logf(l);   // base-2 log-sum-exp
    }
  }
}

}  // namespace

int attention_fwd_bf16(const bf16_t* qkv, bf16_t* out, float* lse, int B, int N, int H, int dh, int nq, hipStream_t st) {
  DGVIT_CHECK_ARG(qkv && out && B > 0 && H > 0, "attention_bf16: bad arguments");
  DGVIT_CHECK_ARG(dh == 64, "attention_bf16: dim_head=%d unsupported (64)", dh);
  DGVIT_CHECK_ARG(N >= 1 && N <= 224, "attention_bf16: tokens N=%d outside [1, 224]", N);
  DGVIT_CHECK_ARG(nq >= 1 && nq <= N, "attention_bf16: bad query limit");
  const int NP = (N + 31) / 32 * 32;
  const size_t lds = (size_t)NP * 128 + (size_t)64 * (NP + 8) * 2;
  const float scale = 1.0f / sqrtf((float)dh);
  const double flops = 4.0 * (double)nq * N * dh * H * B;
  const int slot = profile_begin(PROF_ATTN_FWD, flops, st);
  hipLaunchKernelGGL((attn_fwd_bf16_kernel<4>), dim3((unsigned)((long long)B * H)), dim3(256), lds, st, qkv, out, lse, N, H, scale, nq);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("attn_fwd_bf16_kernel");
  return DGVIT_OK;
}
