// Fused multi-head self-attention for short sequences (N <= 288 tokens: K and V of one (frame, head) stay in the 160 KB LDS), fp32 on
// v_mfma_f32_32x32x2_f32.
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
// Forward walks the key tiles with an online softmax (running max / sum, one 16-register score tile live at a
// time) and stores the base-2 log-sum-exp of every query row.  Backward recomputes the probabilities tile by
// tile from it (only q/k/v/o/lse are saved):
//   phase 1 (wave = query tile, K/V in LDS):  P^T = exp2(S^T - lse), dP^T = V dO^T, dS^T = P^T o (dP^T - delta) * scale,
//                                             dQ^T += K^T dS^T
//   phase 2 (wave = key tile, Q/dO in LDS):   P = exp2(S - lse), dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS
// with delta[q] = sum_d dO[q][d] O[q][d].  Register use is independent of N (occupancy 2+ waves/SIMD for every N).
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// softmax runs in base 2: the query is pre-scaled by scale*log2(e), so exp(x - max) = exp2(s' - max') on v_exp_f32
#define DGVIT_LOG2E 1.4426950408889634f

// Stage rows [0, nrows) of two DH-wide per-head column blocks into LDS images [NP][SK], zero padding rows.
// Every loop trip issues 2*CH float4 loads per thread before it writes LDS (a one-load-per-trip loop would
// serialise a memory round trip per float4).  Out-of-range rows read row 0 and are zeroed by a multiply.
template <int DH, int SK, int NTHR>
__device__ __forceinline__ void stage_pair(float* dstA, const float* srcA, long long ldA, float* dstB, const float* srcB,
                                           long long ldB, int nrows, int NP, int tid) {
  constexpr int C4 = DH / 4, CH = 8;   // 16 float4 in flight per thread: N <= 64 (and N <= 128 with 4 waves) stage in ONE trip
  const int total = NP * C4;
  for (int f0 = tid; f0 < total; f0 += NTHR * CH) {
    float4 va[CH], vb[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int f = f0 + j * NTHR;
      const int row = f / C4, c = (f % C4) * 4;
      const int rr = (f < total && row < nrows) ? row : 0;
      va[j] = *reinterpret_cast<const float4*>(srcA + rr * ldA + c);
      vb[j] = *reinterpret_cast<const float4*>(srcB + rr * ldB + c);
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int f = f0 + j * NTHR;
      const int row = f / C4, c = (f % C4) * 4;
      if (f < total) {
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

template <int DT>
__device__ __forceinline__ void zero_tiles(f32x16 (&t)[DT]) {
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) t[dt][r] = 0.f;
}

// ------------------------------------------------------------------------------------ forward
// NKT_CT > 0: the number of 32-key tiles is a compile-time constant (N <= 64: the loops below unroll into straight-line
// code with no rescale step); NKT_CT == 0: run-time loops for any N
template <int DH, int NW, int NKT_CT>
__global__ void __launch_bounds__(64 * NW, 2) attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                              float* __restrict__ lse, int N, int H, float scale, int nq) {
  constexpr int SK = DH + 4, DT = DH / 32, NTHR = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nkt = NKT_CT ? NKT_CT : (N + 31) / 32, NP = nkt * 32;
  float* Ks = smem;
  float* Vs = smem + NP * SK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int I = H * DH;
  const long long ld = 3ll * I;
  const float* base = qkv + (long long)b * N * ld + hd * DH;
  const float qscale = scale * DGVIT_LOG2E;

  const int nqt = (nq + 31) / 32;   // only queries < nq are needed (nq = 1: the last block keeps token 0 only)
  // the per-lane Q fragments are requested BEFORE the K/V staging so both global round trips overlap
  float4 qf[DH / 8];
  {
    const int q0 = wave * 32 + li;
    row_frags<DH>(qf, base + (q0 < nq ? q0 : 0) * ld, q0 < nq, h, qscale);
  }
  stage_pair<DH, SK, NTHR>(Ks, base + I, ld, Vs, base + 2 * I, ld, N, NP, tid);
  __syncthreads();

  for (int qt = wave; qt < nqt; qt += NW) {
    const int q = qt * 32 + li;
    if (qt != wave) row_frags<DH>(qf, base + (q < nq ? q : 0) * ld, q < nq, h, qscale);
    float m = -INFINITY, l = 0.f;
    f32x16 o[DT];
    zero_tiles<DT>(o);
#pragma unroll NKT_CT ? 2 : 1
    for (int kt = 0; kt < nkt; kt += 2) {
      // two key tiles per trip: both score tiles come out of one back-to-back MFMA batch, one max / rescale serves
      // both, then one MFMA batch for P.V (for N <= 64 this is the whole softmax in a single pass)
      const bool two = kt + 1 < nkt;   // wave-uniform
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = 0.f;
        s1[r] = 0.f;
      }
      mfma_rows_x_frags<DH, SK>(s0, Ks, kt * 32 + li, h, qf);
      if (two) mfma_rows_x_frags<DH, SK>(s1, Ks, (kt + 1) * 32 + li, h, qf);
      float mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + acc_row(r, h);
        const float v0 = key < N ? s0[r] : -INFINITY;
        const float v1 = (two && key + 32 < N) ? s1[r] : -INFINITY;
        s0[r] = v0;
        s1[r] = v1;
        mt = fmaxf(mt, fmaxf(v0, v1));
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float mn = fmaxf(m, mt);                       // every trip holds at least one real key: mn is finite
      const float alpha = __builtin_amdgcn_exp2f(m - mn);  // first trip: exp2(-inf) = 0
      float ts = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p0 = __builtin_amdgcn_exp2f(s0[r] - mn), p1 = __builtin_amdgcn_exp2f(s1[r] - mn);
        s0[r] = p0;
        s1[r] = p1;
        ts += p0 + p1;
      }
      ts += __shfl_xor(ts, 32, 64);
      l = l * alpha + ts;
      m = mn;
      if (kt > 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
      // (a step contracts keys acc_row(r, 0) and + 4 of the tile; past N both probabilities are exact zeros: the step is skipped,
      //  wave-uniformly -- at N = 50, 6 of the second tile's 16 steps)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (kt * 32 + acc_row(r, 0) >= N) continue;
        const float* vrow = Vs + (kt * 32 + acc_row(r, h)) * SK + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[dt * 32], s0[r], o[dt], 0, 0, 0);
      }
      if (two) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if ((kt + 1) * 32 + acc_row(r, 0) >= N) continue;
          const float* vrow = Vs + ((kt + 1) * 32 + acc_row(r, h)) * SK + li;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[dt * 32], s1[r], o[dt], 0, 0, 0);
        }
      }
    }
    if (q < nq) {
      store_T<DH>(o, out + ((long long)b * N + q) * I + hd * DH, h, 1.f / l);
      if (lse && h == 0) lse[((long long)b * H + hd) * N + q] = m + __builtin_amdgcn_logf(l);   // base-2 log-sum-exp
    }
  }
}

// ------------------------------------------------------------------------------------ forward, 32 < N <= 64, many heads: pipelined
// The kernel above is bound by latency, not by its matrix work (MFMA-busy 0.35-0.38 at the C3 shape): every workgroup waits for its
// K / V / Q loads, computes, stores, and the four workgroups a CU holds drift apart only slowly.  Here a workgroup (2 waves, one per
// query tile) walks a strided list of (frame, head) items and keeps the NEXT item's K / V rows (16 float4 per thread) and Q fragments
// in registers while it computes the current one; after the compute they go to the same LDS images (35 KB: still four workgroups per
// CU).  Same arithmetic, same order: bit-identical to attn_fwd_kernel<DH, 2, 2>.
template <int DH>
__global__ void __launch_bounds__(128, 2) attn_fwd_pipe_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                               float* __restrict__ lse, int N, int H, float scale, int items) {
  constexpr int SK = DH + 4, DT = DH / 32, NTHR = 128, NP = 64, C4 = DH / 4, CH = NP * C4 / NTHR;   // CH = 8 float4 per image and thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = smem + NP * SK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int I = H * DH;
  const long long ld = 3ll * I;
  const float qscale = scale * DGVIT_LOG2E;
  const int q = wave * 32 + li;
  const bool qv = q < N;

  float4 ka[CH], va[CH], qf[DH / 8];
  auto fetch_q = [&](int item) {    // the lane's Q fragments of `item` (pre-scaled; rows >= N zero)
    const float* base = qkv + (long long)(item / H) * N * ld + (item % H) * DH;
    row_frags<DH>(qf, base + (qv ? q : 0) * ld, qv, h, qscale);
  };
  // K / V rows of `item` -> registers, 16-byte buffer loads: one VGPR of offset (row tid / 16, columns 4 (tid % 16)), the 8-row steps and
  // the K / V column blocks in the scalar offset; the descriptor ends with the frame's last row, so rows >= N read zeros
  const unsigned voff = ((unsigned)(tid / C4) * (unsigned)ld + (unsigned)(tid % C4) * 4u) * 4u;
  auto fetch = [&](int item) {
    const int b = item / H, hd = item % H;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qkv + (long long)b * N * ld), 0, (int)(N * ld * 4), 0x00020000);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const unsigned so = ((unsigned)(j * (NTHR / C4)) * (unsigned)ld + (unsigned)(I + hd * DH)) * 4u;
      ka[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0));
      va[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so + (unsigned)I * 4u, 0));
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int f = tid + j * NTHR, row = f / C4, c = (f % C4) * 4;
      *reinterpret_cast<float4*>(Ks + row * SK + c) = ka[j];
      *reinterpret_cast<float4*>(Vs + row * SK + c) = va[j];
    }
  };

  int item = blockIdx.x;
  if (item >= items) return;
  fetch(item);
  fetch_q(item);
  while (true) {
    stash();
    __syncthreads();                       // the images of `item` are complete
    const int next = item + gridDim.x;
    if (next < items) fetch(next);         // in flight during the compute below
    // ---- both key tiles in one pass (N <= 64): no rescale step
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = 0.f;
      s1[r] = 0.f;
    }
    mfma_rows_x_frags<DH, SK>(s0, Ks, li, h, qf);
    mfma_rows_x_frags<DH, SK>(s1, Ks, 32 + li, h, qf);
    if (next < items) fetch_q(next);       // the fragment registers are free again: the next item's arrive under the softmax and P V
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = acc_row(r, h);
      const float v0 = key < N ? s0[r] : -INFINITY;
      const float v1 = key + 32 < N ? s1[r] : -INFINITY;
      s0[r] = v0;
      s1[r] = v1;
      mt = fmaxf(mt, fmaxf(v0, v1));
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(-INFINITY, mt);                 // (the general kernel's first trip: m = -inf, alpha = 0, l = 0)
    float ts = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p0 = __builtin_amdgcn_exp2f(s0[r] - mn), p1 = __builtin_amdgcn_exp2f(s1[r] - mn);
      s0[r] = p0;
      s1[r] = p1;
      ts += p0 + p1;
    }
    ts += __shfl_xor(ts, 32, 64);
    const float l = 0.f * __builtin_amdgcn_exp2f(-INFINITY - mn) + ts;
    f32x16 o[DT];
    zero_tiles<DT>(o);
    // P V, eight keys (four k-steps) at a time: their V values are read from LDS as one batch in front of the eight MFMAs (one read and
    // one wait per MFMA pair left the LDS latency exposed sixteen times per tile); a group of eight padding keys is skipped (P = 0)
    auto pv_tile = [&](const f32x16& pt, int k0) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (k0 + 8 * g4 >= N) continue;
        float vv[4][DT];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float* vrow = Vs + (k0 + acc_row(4 * g4 + r4, h)) * SK + li;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) vv[r4][dt] = vrow[dt * 32];
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[r4][dt], pt[4 * g4 + r4], o[dt], 0, 0, 0);
      }
    };
    pv_tile(s0, 0);
    pv_tile(s1, 32);
    if (qv) {
      const int b = item / H, hd = item % H;
      store_T<DH>(o, out + ((long long)b * N + q) * I + hd * DH, h, 1.f / l);
      if (lse && h == 0) lse[((long long)b * H + hd) * N + q] = mn + __builtin_amdgcn_logf(l);
    }
    if (next >= items) break;
    item = next;
    __syncthreads();                       // every wave is done with the images before they are overwritten
  }
}

template <int DH>
int launch_fwd_pipe(const float* qkv, float* out, float* lse, int B, int N, int H, float scale, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * 64 * (DH + 4) * sizeof(float);
  const int items = B * H;
  const int grid = items < 4 * 256 ? items : 4 * 256;      // four resident workgroups per CU
  const int slot = profile_begin(PROF_ATTN_FWD, 4.0 * B * H * (double)N * N * DH, stream);
  hipLaunchKernelGGL(attn_fwd_pipe_kernel<DH>, dim3(grid), dim3(128), lds, stream, qkv, out, lse, N, H, scale, items);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("attention_fwd_pipe");
  return DGVIT_OK;
}

// ------------------------------------------------------------------------------------ backward
template <int DH, int NW, int NKT_CT>
__global__ void __launch_bounds__(64 * NW, 2) attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ o_fwd,
                                                              const float* __restrict__ d_out, const float* __restrict__ lse,
                                                              float* __restrict__ dqkv, int N, int H, float scale, int nq) {
  constexpr int SK = DH + 4, DT = DH / 32, NTHR = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nkt = NKT_CT ? NKT_CT : (N + 31) / 32, NP = nkt * 32;
  float* X = smem;                 // phase 1: K      phase 2: Q
  float* Y = smem + NP * SK;       // phase 1: V      phase 2: dO
  float* lse_s = smem + 2 * NP * SK;
  float* del_s = lse_s + NP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int I = H * DH;
  const long long ld = 3ll * I;
  const float* base = qkv + (long long)b * N * ld + hd * DH;
  const float* obase = o_fwd + (long long)b * N * I + hd * DH;
  const float* dobase = d_out + (long long)b * N * I + hd * DH;
  const float* lbase = lse + ((long long)b * H + hd) * N;
  float* gbase = dqkv + (long long)b * N * ld + hd * DH;
  const float qscale = scale * DGVIT_LOG2E;

  const int nqt = (nq + 31) / 32;
  float4 qf[DH / 8], dof[DH / 8];
  float delta, lq;
  auto load_q = [&](int qt) {   // per-lane fragments of one query tile, delta = rowsum(dO o O), lse of the row
    const int q = qt * 32 + li;
    const bool v = q < nq;
    const int qc = v ? q : 0;
    float4 of[DH / 8];
    row_frags<DH>(qf, base + qc * ld, v, h, qscale);
    row_frags<DH>(dof, dobase + (long long)qc * I, v, h, 1.f);
    row_frags<DH>(of, obase + (long long)qc * I, v, h, 1.f);
    lq = v ? lbase[qc] : 0.f;
    float d = 0.f;
#pragma unroll
    for (int g = 0; g < DH / 8; ++g) d += (dof[g].x * of[g].x + dof[g].y * of[g].y) + (dof[g].z * of[g].z + dof[g].w * of[g].w);
    delta = d + __shfl_xor(d, 32, 64);
  };
  load_q(wave);   // requested before the K/V staging (overlapping round trips)
  stage_pair<DH, SK, NTHR>(X, base + I, ld, Y, base + 2 * I, ld, N, NP, tid);
  __syncthreads();

  // ---- phase 1: one query tile per wave -> dQ; lse / delta of the tile go to LDS for phase 2
  for (int qt = wave; qt < nqt; qt += NW) {
    const int q = qt * 32 + li;
    if (qt != wave) load_q(qt);
    if (h == 0) {
      lse_s[q] = lq;
      del_s[q] = delta;
    }
    f32x16 dq[DT];
    zero_tiles<DT>(dq);
#pragma unroll NKT_CT ? 2 : 1
    for (int kt = 0; kt < nkt; kt += 2) {
      const bool two = kt + 1 < nkt;   // wave-uniform; two key tiles per trip (one MFMA batch, one VALU block, one MFMA batch)
      f32x16 s0, s1, dp0, dp1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = 0.f; s1[r] = 0.f; dp0[r] = 0.f; dp1[r] = 0.f;
      }
      mfma_rows_x_frags<DH, SK>(s0, X, kt * 32 + li, h, qf);     // S^T (base-2 scaled)
      mfma_rows_x_frags<DH, SK>(dp0, Y, kt * 32 + li, h, dof);   // dP^T[key][q] = sum_d V[key][d] dO[q][d]
      if (two) {
        mfma_rows_x_frags<DH, SK>(s1, X, (kt + 1) * 32 + li, h, qf);
        mfma_rows_x_frags<DH, SK>(dp1, Y, (kt + 1) * 32 + li, h, dof);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + acc_row(r, h);
        const float p0 = key < N ? __builtin_amdgcn_exp2f(s0[r] - lq) : 0.f;
        const float p1 = (two && key + 32 < N) ? __builtin_amdgcn_exp2f(s1[r] - lq) : 0.f;
        s0[r] = p0 * (dp0[r] - delta) * scale;                   // dS^T
        s1[r] = p1 * (dp1[r] - delta) * scale;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* krow = X + (kt * 32 + acc_row(r, h)) * SK + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[dt * 32], s0[r], dq[dt], 0, 0, 0);
      }
      if (two) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* krow = X + ((kt + 1) * 32 + acc_row(r, h)) * SK + li;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[dt * 32], s1[r], dq[dt], 0, 0, 0);
        }
      }
    }
    if (q < nq) store_T<DH>(dq, gbase + q * ld, h, 1.f);
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
  stage_pair<DH, SK, NTHR>(X, base, ld, Y, dobase, (long long)I, nq, NP, tid);   // rows >= nq zero-filled: no gradient
  __syncthreads();
  for (int kt = wave; kt < nkt; kt += NW) {
    const int key = kt * 32 + li;
    const bool kv = key < N;
    if (kt != wave) {
      const int kc = kv ? key : 0;
      row_frags<DH>(kf, base + I + kc * ld, kv, h, 1.f);
      row_frags<DH>(vf, base + 2 * I + kc * ld, kv, h, 1.f);
    }
    f32x16 dk[DT], dv[DT];
    zero_tiles<DT>(dk);
    zero_tiles<DT>(dv);
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
        const float pv = (kv && q < nq) ? __builtin_amdgcn_exp2f(s[r] * qscale - lse_s[q]) : 0.f;
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

// ------------------------------------------------------------------------------------ backward, 32 < N <= 64, all queries
// (the DGViT-small shapes: N = 50 at 84x84 @ 12x12, N = 37 @ 14x14).  The two-phase kernel above recomputes S and dP in the second
// orientation: 224 MFMAs per (query tile, key tile) pair.  Here every pair is computed ONCE, by its own wave (4 waves = 2 query
// tiles x 2 key tiles):
//   step 1 (wave = (qt, kt), K / V images in LDS, query on the lane): S^T, dP^T -> P^T, dS^T in registers; both tiles also go to
//          LDS as [key][query] images; dQ^T partial = K^T dS^T straight from the registers.           96 MFMAs
//   step 2 the two key-tile partials of dQ are added through LDS (fixed order) and stored.
//   step 3 (Q / dO images written over K / V from the fragments step 1 already holds; wave = (kt, qt), key on the lane): the [key][query] images are read back as B
//          operands (one float4 per 4 queries): dV^T += dO^T P, dK^T += Q^T dS.                           64 MFMAs
//   step 4 the two query-tile partials of dK / dV are added through LDS and stored.
// 160 MFMAs per pair instead of 224, no second fetch of K / V fragments, no lse / delta arrays; 70 KB of LDS (2 workgroups per CU,
// 8 waves, as before) and a third of the per-wave dependent MFMA chain.
template <int DH>
__global__ void __launch_bounds__(256, 3) attn_bwd64_kernel(const float* __restrict__ qkv, const float* __restrict__ o_fwd,
                                                            const float* __restrict__ d_out, const float* __restrict__ lse,
                                                            float* __restrict__ dqkv, int N, int H, float scale) {
  constexpr int SK = DH + 4, DT = DH / 32, NP = 64, TQ = NP + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X = smem;                 // step 1: K      step 3: Q          (step 4: dV partial [key][d])
  float* Y = X + NP * SK;          // step 1: V      step 3: dO
  float* T = Y + NP * SK;          // [key][query] image: P^T for dV, then dS^T for dK   (step 4: dK partial [key][d])
  // (ONE [key][query] image, used twice -- P^T, then dS^T written from the registers that kept it -- instead of two: 52 KB of LDS
  //  per workgroup instead of 70, three workgroups per CU instead of two; the kernel is bound by latency, not by its matrix work)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int I = H * DH;
  const long long ld = 3ll * I;
  const float* base = qkv + (long long)b * N * ld + hd * DH;
  const float* obase = o_fwd + (long long)b * N * I + hd * DH;
  const float* dobase = d_out + (long long)b * N * I + hd * DH;
  float* gbase = dqkv + (long long)b * N * ld + hd * DH;
  const float qscale = scale * DGVIT_LOG2E;
  f32x16 dsT;                      // dS^T tile of this wave's (query tile, key tile) pair, kept from step 1 to step 3b
  const int qt1 = wave & 1, kt1 = wave >> 1, q1 = qt1 * 32 + li;
  // ---- step 1: wave (qt1, kt1)
  {
    const int qt = qt1, kt = kt1, q = q1;
    const bool qv = q < N;
    const int qc = qv ? q : 0;
    float4 qf[DH / 8], dof[DH / 8];
    float delta, lq;
    {
      // EVERY global load of the workgroup is issued before the first one is waited for -- the K / V staging pieces (NP * DH / 4 / 256
      // float4 per thread and matrix, rows >= N read row 0 and are zeroed), then the q / dO / O row fragments and the row's lse: one
      // memory round trip.  (stage_pair() after the delta sum cost a second one: the sum waits for the fragments, and its loop issued
      // half of its loads for pieces past the image.)
      constexpr int C4 = DH / 4, PER = NP * C4 / 256;
      float4 kst[PER], vst[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int f = tid + j * 256, row = f / C4, c = (f % C4) * 4;
        const int rr = row < N ? row : 0;
        kst[j] = *reinterpret_cast<const float4*>(base + I + rr * ld + c);
        vst[j] = *reinterpret_cast<const float4*>(base + 2 * I + rr * ld + c);
      }
      float4 of[DH / 8];
      row_frags<DH>(qf, base + qc * ld, qv, h, qscale);
      row_frags<DH>(dof, dobase + (long long)qc * I, qv, h, 1.f);
      row_frags<DH>(of, obase + (long long)qc * I, qv, h, 1.f);
      lq = qv ? lse[((long long)b * H + hd) * N + qc] : 0.f;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int f = tid + j * 256, row = f / C4, c = (f % C4) * 4;
        const float k = row < N ? 1.f : 0.f;   // (a float4 ?: would be lowered through scratch memory)
        *reinterpret_cast<float4*>(X + row * SK + c) = make_float4(kst[j].x * k, kst[j].y * k, kst[j].z * k, kst[j].w * k);
        *reinterpret_cast<float4*>(Y + row * SK + c) = make_float4(vst[j].x * k, vst[j].y * k, vst[j].z * k, vst[j].w * k);
      }
      float d = 0.f;
#pragma unroll
      for (int g = 0; g < DH / 8; ++g) d += (dof[g].x * of[g].x + dof[g].y * of[g].y) + (dof[g].z * of[g].z + dof[g].w * of[g].w);
      delta = d + __shfl_xor(d, 32, 64);
    }
    __syncthreads();
    f32x16 sT, dpT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sT[r] = 0.f;
      dpT[r] = 0.f;
    }
    mfma_rows_x_frags<DH, SK>(sT, X, kt * 32 + li, h, qf);      // S^T[key][query] (base-2 scaled)
    mfma_rows_x_frags<DH, SK>(dpT, Y, kt * 32 + li, h, dof);    // dP^T[key][query]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + acc_row(r, h);
      const float pv = key < N ? __builtin_amdgcn_exp2f(sT[r] - lq) : 0.f;
      T[key * TQ + q] = pv;          // consecutive lanes -> consecutive queries: conflict-free
      dsT[r] = pv * (dpT[r] - delta) * scale;
    }
    f32x16 dq[DT];
    zero_tiles<DT>(dq);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      // four k-steps = eight keys at a time, their K values read from LDS as one batch in front of the eight MFMAs (a read and a wait per
      // MFMA pair left the LDS latency exposed sixteen times); a group of eight padding keys (dS = 0) is skipped, wave-uniform
      if (kt * 32 + 8 * g4 >= N) continue;
      float kv[4][DT];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float* krow = X + (kt * 32 + acc_row(4 * g4 + r4, h)) * SK + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) kv[r4][dt] = krow[dt * 32];
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[r4][dt], dsT[4 * g4 + r4], dq[dt], 0, 0, 0);
    }
    __syncthreads();                 // every wave is done with the K / V images
    // ---- step 2: dQ = partial(kt = 0) + partial(kt = 1), through the (now free) X image, rows = queries
    float* ex = X + (qt * 32 + li) * SK;
    if (kt == 1) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          *reinterpret_cast<float4*>(ex + dt * 32 + 8 * c + 4 * h) = make_float4(dq[dt][4 * c], dq[dt][4 * c + 1], dq[dt][4 * c + 2], dq[dt][4 * c + 3]);
    }
    __syncthreads();
    if (kt == 0 && qv) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 o = *reinterpret_cast<const float4*>(ex + dt * 32 + 8 * c + 4 * h);
          *reinterpret_cast<float4*>(gbase + q * ld + dt * 32 + 8 * c + 4 * h) =
              make_float4(dq[dt][4 * c] + o.x, dq[dt][4 * c + 1] + o.y, dq[dt][4 * c + 2] + o.z, dq[dt][4 * c + 3] + o.w);
        }
    }
    __syncthreads();
    // ---- step 3 images straight from the registers: the kt = 0 waves hold every query row (q pre-scaled, undone at the store) and
    // every dO row as fragments already; rows >= N are zero.  No second trip to global memory.
    if (kt == 0) {
#pragma unroll
      for (int g = 0; g < DH / 8; ++g) {
        *reinterpret_cast<float4*>(X + q * SK + 8 * g + 4 * h) = qf[g];
        *reinterpret_cast<float4*>(Y + q * SK + 8 * g + 4 * h) = dof[g];
      }
    }
    __syncthreads();
  }
  // ---- step 3: wave (kt, qt), key on the lane.  3a: dV^T += dO^T P from the P^T image; then the image is rewritten with dS^T
  // (each wave its step-1 tile) for 3b: dK^T += Q^T dS
  {
    const int kt = wave & 1, qt = wave >> 1;
    const int key = kt * 32 + li;
    f32x16 dk[DT], dv[DT];
    zero_tiles<DT>(dk);
    zero_tiles<DT>(dv);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (qt * 32 + 8 * g >= N) continue;             // eight padding queries (dO = 0): wave-uniform skip
      const float4 pf = *reinterpret_cast<const float4*>(T + key * TQ + qt * 32 + 8 * g + 4 * h);
      const float pe[4] = {pf.x, pf.y, pf.z, pf.w};
      float dov[4][DT];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* dorow = Y + (qt * 32 + 8 * g + 4 * h + e) * SK + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dov[e][dt] = dorow[dt * 32];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dov[e][dt], pe[e], dv[dt], 0, 0, 0);
    }
    __syncthreads();                 // P^T has been consumed
#pragma unroll
    for (int r = 0; r < 16; ++r) T[(kt1 * 32 + acc_row(r, h)) * TQ + q1] = dsT[r];
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (qt * 32 + 8 * g >= N) continue;             // (dS = 0)
      const float4 sf = *reinterpret_cast<const float4*>(T + key * TQ + qt * 32 + 8 * g + 4 * h);
      const float se[4] = {sf.x, sf.y, sf.z, sf.w};
      float qv4[4][DT];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* qrow = X + (qt * 32 + 8 * g + 4 * h + e) * SK + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) qv4[e][dt] = qrow[dt * 32];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv4[e][dt], se[e], dk[dt], 0, 0, 0);
    }
    __syncthreads();                 // the dS^T image and the Q / dO images have been consumed
    // ---- step 4: dK / dV = partial(qt = 0) + partial(qt = 1), through the T / X regions, rows = keys
    float* ek = T + key * TQ;
    float* ev = X + key * SK;
    if (qt == 1) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          *reinterpret_cast<float4*>(ek + dt * 32 + 8 * c + 4 * h) = make_float4(dk[dt][4 * c], dk[dt][4 * c + 1], dk[dt][4 * c + 2], dk[dt][4 * c + 3]);
          *reinterpret_cast<float4*>(ev + dt * 32 + 8 * c + 4 * h) = make_float4(dv[dt][4 * c], dv[dt][4 * c + 1], dv[dt][4 * c + 2], dv[dt][4 * c + 3]);
        }
    }
    __syncthreads();
    if (qt == 0 && key < N) {
      const float unscale = 1.f / qscale;   // the Q image holds q * scale * log2(e)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 a = *reinterpret_cast<const float4*>(ek + dt * 32 + 8 * c + 4 * h);
          const float4 bq = *reinterpret_cast<const float4*>(ev + dt * 32 + 8 * c + 4 * h);
          *reinterpret_cast<float4*>(gbase + I + key * ld + dt * 32 + 8 * c + 4 * h) =
              make_float4((dk[dt][4 * c] + a.x) * unscale, (dk[dt][4 * c + 1] + a.y) * unscale, (dk[dt][4 * c + 2] + a.z) * unscale,
                          (dk[dt][4 * c + 3] + a.w) * unscale);
          *reinterpret_cast<float4*>(gbase + 2 * I + key * ld + dt * 32 + 8 * c + 4 * h) =
              make_float4(dv[dt][4 * c] + bq.x, dv[dt][4 * c + 1] + bq.y, dv[dt][4 * c + 2] + bq.z, dv[dt][4 * c + 3] + bq.w);
        }
    }
  }
}


template <int DH>
int launch_bwd64(const float* qkv, const float* o, const float* dout, const float* lse, float* dqkv, int B, int N, int H, float scale,
                 hipStream_t stream) {
  constexpr size_t lds = (size_t)(2 * 64 * (DH + 4) + 64 * 68) * sizeof(float);   // 52 KB at dim_head 64: three workgroups per CU
  auto kern = attn_bwd64_kernel<DH>;
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, "attention_bwd: %s", hipGetErrorString(e));
    once.mark(bit);
  }
  const int slot = profile_begin(PROF_ATTN_BWD, 8.0 * B * H * (double)N * N * DH, stream);
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(256), lds, stream, qkv, o, dout, lse, dqkv, N, H, scale);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("attention_bwd64");
  return DGVIT_OK;
}

// ------------------------------------------------------------------------------------ ONE query (token 0), N <= 64: no matrix cores
// The last block of the encoder keeps token 0 only (GoalFormer.py:167 reads x[:, 0]): its attention has one query row per (frame, head).
// On the tile kernels above that row is 1 of 32 rows of an MFMA tile and the launch is all latency (forward 49 us, backward 124 us at the
// C3 shape for 105 / 210 MB of K, V, dK, dV).  Here a WAVE takes one (frame, head): DH / 4 adjacent lanes hold one key's K and V rows as
// float4 pieces (whole 128 / 256-byte row segments per load instruction, 64 * 4 / DH keys per instruction, every load of the wave issued
// before the first is used), the dot products are DPP butterflies over those lanes, and the softmax / weighted sums are a handful of FMAs.
// Plain fp32 FMAs: not bit-identical to the MFMA kernels' summation order, same 1e-6-level agreement as between any two of them.
template <int W>
__device__ __forceinline__ float lanes_sum(float v) {     // over W adjacent lanes (butterfly: every lane ends with the sum)
#pragma unroll
  for (int o = W >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int W>
__device__ __forceinline__ float groups_sum(float v) {    // over the 64 / W groups of W lanes
#pragma unroll
  for (int o = W; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

template <int DH>
__global__ void __launch_bounds__(256) attn_q1_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                          int N, int H, float scale, int items) {
  constexpr int C4 = DH / 4, G = 64 / C4, NI = 64 / G;     // lanes per key, keys per step, steps (64 keys)
  const int lane = threadIdx.x & 63, item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= items) return;                                // (wave-uniform; no barriers in this kernel)
  const int b = item / H, hd = item - b * H, I = H * DH;
  const long long ld = 3ll * I;
  const int c = lane % C4, g = lane / C4;
  const float* base = qkv + (long long)b * N * ld + hd * DH + 4 * c;
  float4 kf[NI], vf[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int key = i * G + g, rr = key < N ? key : 0;
    kf[i] = *reinterpret_cast<const float4*>(base + I + rr * ld);
    vf[i] = *reinterpret_cast<const float4*>(base + 2 * I + rr * ld);
  }
  float4 q = *reinterpret_cast<const float4*>(base);
  const float qscale = scale * DGVIT_LOG2E;
  q = make_float4(q.x * qscale, q.y * qscale, q.z * qscale, q.w * qscale);
  float sc[NI], m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const float d = lanes_sum<C4>(dot4(q, kf[i]));
    sc[i] = i * G + g < N ? d : -INFINITY;
    m = fmaxf(m, sc[i]);
  }
#pragma unroll
  for (int o = C4; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));     // (key 0 is real: m is finite)
  float l = 0.f;
  float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const float pv = __builtin_amdgcn_exp2f(sc[i] - m);     // exp2(-inf) = 0 for the padding keys
    l += pv;
    o4.x += pv * vf[i].x; o4.y += pv * vf[i].y; o4.z += pv * vf[i].z; o4.w += pv * vf[i].w;
  }
  l = groups_sum<C4>(l);
  o4 = make_float4(groups_sum<C4>(o4.x), groups_sum<C4>(o4.y), groups_sum<C4>(o4.z), groups_sum<C4>(o4.w));
  if (g == 0) {
    const float inv = 1.f / l;
    *reinterpret_cast<float4*>(out + (long long)b * N * I + hd * DH + 4 * c) = make_float4(o4.x * inv, o4.y * inv, o4.z * inv, o4.w * inv);
  }
  if (lse && lane == 0) lse[((long long)b * H + hd) * N] = m + __builtin_amdgcn_logf(l);   // base-2 log-sum-exp
}

template <int DH>
__global__ void __launch_bounds__(256) attn_q1_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ o_fwd,
                                                          const float* __restrict__ d_out, const float* __restrict__ lse,
                                                          float* __restrict__ dqkv, int N, int H, float scale, int items) {
  constexpr int C4 = DH / 4, G = 64 / C4, NI = 64 / G;
  const int lane = threadIdx.x & 63, item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= items) return;
  const int b = item / H, hd = item - b * H, I = H * DH;
  const long long ld = 3ll * I;
  const int c = lane % C4, g = lane / C4;
  const float* base = qkv + (long long)b * N * ld + hd * DH + 4 * c;
  float* gbase = dqkv + (long long)b * N * ld + hd * DH + 4 * c;
  float4 kf[NI], vf[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int key = i * G + g, rr = key < N ? key : 0;
    kf[i] = *reinterpret_cast<const float4*>(base + I + rr * ld);
    vf[i] = *reinterpret_cast<const float4*>(base + 2 * I + rr * ld);
  }
  const float4 q = *reinterpret_cast<const float4*>(base);
  const float4 dof = *reinterpret_cast<const float4*>(d_out + (long long)b * N * I + hd * DH + 4 * c);
  const float4 of = *reinterpret_cast<const float4*>(o_fwd + (long long)b * N * I + hd * DH + 4 * c);
  const float lq = lse[((long long)b * H + hd) * N];
  const float qscale = scale * DGVIT_LOG2E;
  const float4 qs = make_float4(q.x * qscale, q.y * qscale, q.z * qscale, q.w * qscale);
  const float delta = lanes_sum<C4>(dot4(dof, of));
  float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int key = i * G + g;
    const float sv = lanes_sum<C4>(dot4(qs, kf[i])), dp = lanes_sum<C4>(dot4(dof, vf[i]));
    const float pv = key < N ? __builtin_amdgcn_exp2f(sv - lq) : 0.f;
    const float ds = pv * (dp - delta) * scale;
    if (key < N) {
      *reinterpret_cast<float4*>(gbase + I + key * ld) = make_float4(ds * q.x, ds * q.y, ds * q.z, ds * q.w);
      *reinterpret_cast<float4*>(gbase + 2 * I + key * ld) = make_float4(pv * dof.x, pv * dof.y, pv * dof.z, pv * dof.w);
    }
    dq.x += ds * kf[i].x; dq.y += ds * kf[i].y; dq.z += ds * kf[i].z; dq.w += ds * kf[i].w;
  }
  dq = make_float4(groups_sum<C4>(dq.x), groups_sum<C4>(dq.y), groups_sum<C4>(dq.z), groups_sum<C4>(dq.w));
  if (g == 0) *reinterpret_cast<float4*>(gbase) = dq;
}

template <int DH>
int launch_q1_fwd(const float* qkv, float* out, float* lse, int B, int N, int H, float scale, hipStream_t stream) {
  const int items = B * H;
  const int slot = profile_begin(PROF_ATTN_FWD, 4.0 * items * (double)N * DH, stream);
  hipLaunchKernelGGL(attn_q1_fwd_kernel<DH>, dim3((items + 3) / 4), dim3(256), 0, stream, qkv, out, lse, N, H, scale, items);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("attention_fwd (one query)");
  return DGVIT_OK;
}

template <int DH>
int launch_q1_bwd(const float* qkv, const float* o, const float* dout, const float* lse, float* dqkv, int B, int N, int H, float scale,
                  hipStream_t stream) {
  const int items = B * H;
  const int slot = profile_begin(PROF_ATTN_BWD, 8.0 * items * (double)N * DH, stream);
  hipLaunchKernelGGL(attn_q1_bwd_kernel<DH>, dim3((items + 3) / 4), dim3(256), 0, stream, qkv, o, dout, lse, dqkv, N, H, scale, items);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("attention_bwd (one query)");
  return DGVIT_OK;
}

constexpr int MAX_TOKENS = 288;   // 2 * 288 * 68 floats (K, V images at dim_head 64) + 2 * 288 (lse, delta) = 159.0 KB of the 160 KB LDS

template <int DH, int NW, int NKT_CT>
int launch_fwd(const float* qkv, float* out, float* lse, int B, int N, int H, float scale, int nq, hipStream_t stream) {
  const int NP = (N + 31) / 32 * 32;
  const size_t lds = (size_t)2 * NP * (DH + 4) * sizeof(float);
  auto kern = attn_fwd_kernel<DH, NW, NKT_CT>;
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    const int maxlds = 2 * MAX_TOKENS * (DH + 4) * (int)sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
    if (e != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, "attention_fwd: %s", hipGetErrorString(e));
    once.mark(bit);
  }
  const int slot = profile_begin(PROF_ATTN_FWD, 4.0 * B * H * (double)nq * N * DH, stream);
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(64 * NW), lds, stream, qkv, out, lse, N, H, scale, nq);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("attention_fwd");
  return DGVIT_OK;
}

template <int DH, int NW, int NKT_CT>
int launch_bwd(const float* qkv, const float* o, const float* dout, const float* lse, float* dqkv, int B, int N, int H,
               float scale, int nq, hipStream_t stream) {
  const int NP = (N + 31) / 32 * 32;
  const size_t lds = ((size_t)2 * NP * (DH + 4) + 2 * NP) * sizeof(float);
  auto kern = attn_bwd_kernel<DH, NW, NKT_CT>;
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    const int maxlds = (2 * MAX_TOKENS * (DH + 4) + 2 * MAX_TOKENS) * (int)sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
    if (e != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, "attention_bwd: %s", hipGetErrorString(e));
    once.mark(bit);
  }
  const int slot = profile_begin(PROF_ATTN_BWD, 8.0 * B * H * (double)nq * N * DH, stream);
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(64 * NW), lds, stream, qkv, o, dout, lse, dqkv, N, H, scale, nq);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("attention_bwd");
  return DGVIT_OK;
}

}  // namespace

// one wave per 32-token tile, at most 4 waves per workgroup; N <= 64 takes the compile-time-unrolled instantiations
#define ATTN_DISPATCH(FN, ...)                                            \
  {                                                                       \
    const int nt = (N + 31) / 32;                                         \
    if (dh == 64 && nt == 1) return FN<64, 1, 1>(__VA_ARGS__);            \
    if (dh == 64 && nt == 2) return FN<64, 2, 2>(__VA_ARGS__);            \
    if (dh == 64) return FN<64, 4, 0>(__VA_ARGS__);                       \
    if (dh == 32 && nt == 1) return FN<32, 1, 1>(__VA_ARGS__);            \
    if (dh == 32 && nt == 2) return FN<32, 2, 2>(__VA_ARGS__);            \
    if (dh == 32) return FN<32, 4, 0>(__VA_ARGS__);                       \
  }

// qkv (B, N, 3*H*dh) -> out (B, N, H*dh); lse (B, H, N) base-2 log-sum-exp of the scaled scores (NULL: not kept).
// nq = number of leading query tokens whose output is needed (N normally)
int attention_fwd(const float* qkv, float* out, float* lse, int B, int N, int H, int dh, int nq, hipStream_t stream) {
  DGVIT_CHECK_ARG(qkv && out && B > 0 && N > 0 && H > 0, "attention_fwd: bad arguments");
  DGVIT_CHECK_ARG((long long)B * H < (1ll << 31), "attention_fwd: B*H too large");
  DGVIT_CHECK_ARG(nq >= 1 && nq <= N, "attention_fwd: bad query count");
  DGVIT_CHECK_ARG(N <= MAX_TOKENS && (dh == 64 || dh == 32), "attention_fwd: unsupported dim_head=%d / tokens=%d (dim_head 64 or 32, N <= 288)", dh, N);
  const float scale = 1.0f / sqrtf((float)dh);
  if (g_attn_q1 && nq == 1 && N <= 64) {                               // token 0 only (the last block): one wave per (frame, head)
    if (dh == 64) return launch_q1_fwd<64>(qkv, out, lse, B, N, H, scale, stream);
    return launch_q1_fwd<32>(qkv, out, lse, B, N, H, scale, stream);
  }
  if (nq == N && N > 32 && N <= 64 && (long long)B * H >= 2048) {      // many short sequences: the pipelined form
    if (dh == 64) return launch_fwd_pipe<64>(qkv, out, lse, B, N, H, scale, stream);
    return launch_fwd_pipe<32>(qkv, out, lse, B, N, H, scale, stream);
  }
  ATTN_DISPATCH(launch_fwd, qkv, out, lse, B, N, H, scale, nq, stream)
  return dgvit_set_error(DGVIT_ERR_ARG, "attention_fwd: no kernel for dim_head=%d", dh);
}

// dqkv (B, N, 3*H*dh); with nq < N only rows < nq of `o`/`dout`/`lse` are read and only rows < nq of dq are written
// (dk, dv: all rows)

int attention_bwd(const float* qkv, const float* o, const float* dout, const float* lse, float* dqkv, int B, int N, int H,
                  int dh, int nq, hipStream_t stream) {
  DGVIT_CHECK_ARG(qkv && o && dout && lse && dqkv && B > 0 && N > 0 && H > 0, "attention_bwd: bad arguments");
  DGVIT_CHECK_ARG((long long)B * H < (1ll << 31), "attention_bwd: B*H too large");
  DGVIT_CHECK_ARG(nq >= 1 && nq <= N, "attention_bwd: bad query count");
  DGVIT_CHECK_ARG(N <= MAX_TOKENS && (dh == 64 || dh == 32), "attention_bwd: unsupported dim_head=%d / tokens=%d", dh, N);
  const float scale = 1.0f / sqrtf((float)dh);
  if (g_attn_q1 && nq == 1 && N <= 64) {
    if (dh == 64) return launch_q1_bwd<64>(qkv, o, dout, lse, dqkv, B, N, H, scale, stream);
    return launch_q1_bwd<32>(qkv, o, dout, lse, dqkv, B, N, H, scale, stream);
  }
  if (g_attn_bwd64 && nq == N && N > 32 && N <= 64) {   // the DGViT-small token counts: every tile pair computed once
    if (dh == 64) return launch_bwd64<64>(qkv, o, dout, lse, dqkv, B, N, H, scale, stream);
    return launch_bwd64<32>(qkv, o, dout, lse, dqkv, B, N, H, scale, stream);
  }
  ATTN_DISPATCH(launch_bwd, qkv, o, dout, lse, dqkv, B, N, H, scale, nq, stream)
  return dgvit_set_error(DGVIT_ERR_ARG, "attention_bwd: no kernel for dim_head=%d", dh);
}
