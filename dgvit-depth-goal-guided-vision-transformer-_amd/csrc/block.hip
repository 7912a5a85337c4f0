// Small-batch forward of the GoT transformer blocks (GoalFormer.py:101-105 with :31-82): TWO launches per block, each with its
// cross-workgroup sum taken INSIDE the launch -- the regime the reference's SAC loop lives in: SAC.choose_action on one frame
// (DRL.py:170-185) and the no-grad passes of learn() at the shipped batch of 32 (config.yaml:11).  Inference only (nothing is kept
// for a backward).
//
// At T = B * N = 65 ... 4160 token rows the seven-launch schedule of dgvit_api.hip is a chain of 5-8 us kernels (launch boundary, one
// or two dependent L2 round trips, a few hundred MFMAs, a split-K hand-off): 0.18 ms for the shipped four-block actor on one frame.
// Here a block is
//
//   attn_block_kernel   one workgroup per (frame b, head h, 32-query tile qt)
//       ln      = LayerNorm1 rows of frame b                        (written by the previous kernel's combine step)
//       k, v    = ln Wk_h^T, ln Wv_h^T   for all N rows;  q = ln Wq_h^T for the 32 rows of qt      (MFMA, weights straight from L2)
//       ao      = softmax(q k^T dh^-1/2) v                          one wave per 32-key tile, scores transposed (keys in registers)
//       part    = ao Wout[:, h dh : (h+1) dh]^T                     this head's share of to_out, 32 x D
//       combine : the LAST of the H workgroups of (b, qt) to arrive sums the shares in head order, adds bias + residual -> xmid,
//                 and normalises the rows (LayerNorm2) -> ln2
//   mlp_block_kernel    one workgroup per (32-row tile rt, chunk c of 128 hidden units)         -- the "fused no-grad MLP"
//       a       = gelu(ln2 W1_c^T + b1_c)            32 x 128, stays in LDS: the hidden layer never exists in HBM
//       part    = a W2[:, c 128 : (c+1) 128]^T       32 x D
//       combine : the last of the M / 128 workgroups of rt sums the shares in chunk order, adds bias + residual -> xout, and
//                 normalises the rows with the NEXT block's LayerNorm1 -> ln1
//
// Redundant work instead of synchronisation: every (b, h, qt) workgroup projects K and V of the whole frame (the chip is empty at
// these sizes; sharing them would cost a hand-off), and the hand-offs that remain are the split-K recipe of gemm.hip: write-through
// (sc1) 16-byte partial stores, every wave drains, barrier, ONE relaxed agent-scope ticket, the last arriver acquires (agent scope)
// and reads the others' partials -- placement independent, no spinning, no residency requirement (cdna_hip_programming.md,
// Guideline 16, R1).  Sums are taken in a fixed order: results are bit-reproducible.
// The last block under the token-0 schedule runs the query tile of token 0 only and feeds the MLP kernel the B
// token-0 rows (row stride N).
// Against frame.hip (round 2, two launches per block with the sums taken by the NEXT kernel, measured slower than seven launches):
// query tiles and 32-row tiles instead of whole frames (3 x the workgroups at N = 65), one wave per key tile in the attention
// instead of one wave per query tile, every weight fragment of a phase requested before its first MFMA and the next batch under
// the current one's MFMAs, LayerNorm once per row in the combine step instead of once per consumer workgroup.
#include "common.h"
#include "kernels.h"
#include "small_mma.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int BHC = 128;            // hidden columns per mlp_block workgroup
#define DGVIT_LOG2E_F 1.4426950408889634f

// acc(32 x 32) += A W^T over k in [0, K): A rows in an LDS image (row = lane & 31, k contiguous, stride sa floats), W row of this lane
// from global memory (k contiguous).  K % 8 == 0.  Weight fragments travel in batches of four 8-deep k-groups (4 x 16 bytes per lane);
// the next batch is requested before the MFMAs of the current one, so one round trip is exposed per call, not one per batch.
__device__ __forceinline__ void mm_lds_x_wrow(f32x16& acc, const float* arow_, const float* __restrict__ wrow, int K, int h) {
  const int ng = K >> 3;
  float4 b0[4], b1[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) b0[u] = u < ng ? *reinterpret_cast<const float4*>(wrow + 8 * u + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g0 = 0; g0 < ng; g0 += 8) {
#pragma unroll
    for (int u = 0; u < 4; ++u) b1[u] = g0 + 4 + u < ng ? *reinterpret_cast<const float4*>(wrow + 8 * (g0 + 4 + u) + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (g0 + u < ng) {
        const float4 a = *reinterpret_cast<const float4*>(arow_ + 8 * (g0 + u) + 4 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0[u].w, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) b0[u] = g0 + 8 + u < ng ? *reinterpret_cast<const float4*>(wrow + 8 * (g0 + 8 + u) + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (g0 + 4 + u < ng) {
        const float4 a = *reinterpret_cast<const float4*>(arow_ + 8 * (g0 + 4 + u) + 4 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1[u].w, acc, 0, 0, 0);
      }
    }
  }
}

// part[ks][32][D + 4] (LDS) = k-slices of  A (32 x K, LDS rows a_img, stride sa)  x  W[j][koff + k]^T  for j in [0, D): the D / 32 column
// tiles are dealt over the four waves; with fewer than four tiles (D = 64, 32) the waves also split K (ks = 4 / tiles slices, summed
// by the reader in slice order).  Returns ks.
__host__ __device__ __forceinline__ int out_ksplit(int D) { return D >= 128 ? 1 : 128 / D; }   // D in {32, 64, 128, 256}: 4, 2, 1, 1

__device__ __forceinline__ int rows32_x_w_to_lds(float* part, const float* a_img, int sa, const float* __restrict__ W, int ldw, int koff, int D,
                                                 int K, int wave, int li, int h) {
  const int nt = D >> 5, SP = D + 4;
  const int ks = out_ksplit(D);
  const int kslice = K / ks;
  for (int t = nt >= 4 ? wave : wave % nt; t < nt; t += 4) {
    const int kp = nt >= 4 ? 0 : wave / nt;
    f32x16 acc;
    zero16(acc);
    mm_lds_x_wrow(acc, a_img + li * sa + kp * kslice, W + (long long)(t * 32 + li) * ldw + koff + kp * kslice, kslice, h);
    float* dst = part + (long long)kp * 32 * SP + t * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[arow(r, h) * SP] = acc[r];
  }
  return ks;
}

struct CombineArgs {
  float* slabs;              // [group][nparts][32][D]   partial rows of every workgroup of a group
  int* counters;             // [group]                  zero on entry, left zero
  const float* bias;         // (D)
  const float* res; float* out;          // residual rows in, finished rows out (row stride ld floats each)
  const float* lnw; const float* lnb;    // LayerNorm applied to the finished rows, or null
  float* ln_out;                         // its output (row stride ld)
  long long ld;
};

// The epilogue both kernels share.  `part` = this workgroup's 32 x D partial (ks k-slices) in LDS; `group` = the set of workgroups
// whose partials add up to the same 32 rows, `slot` = this workgroup's place in the fixed summation order, `nparts` = its size;
// row r of the tile is global row row0 + r * rstep (rows with r >= nrows do not exist).  flag: one int of LDS.
__device__ __forceinline__ void publish_and_combine(const CombineArgs& c, const float* part, int ks, int D, int group, int slot, int nparts,
                                                    long long row0, long long rstep, int nrows, int* flag, int tid) {
  const int SP = D + 4, C4 = D >> 2, RPP = 256 / C4;
  const int cc = (tid % C4) * 4, rr0 = tid / C4;
  float* slab = c.slabs + ((long long)group * nparts + slot) * 32 * D;
  if (nparts > 1) {
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, 32 * D * 4, 0x00020000);
    for (int rr = rr0; rr < 32; rr += RPP) {
      float4 v = *reinterpret_cast<const float4*>(part + rr * SP + cc);
      for (int s = 1; s < ks; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(part + (s * 32 + rr) * SP + cc);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srs, (unsigned)((rr * D + cc) * 4), 0, 16);   // sc1: write-through
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(c.counters + group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == nparts - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(c.counters + group, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
      }
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
  }
  const float4 bias = *reinterpret_cast<const float4*>(c.bias + cc);
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f), be = g;
  if (c.lnw) {
    g = *reinterpret_cast<const float4*>(c.lnw + cc);
    be = *reinterpret_cast<const float4*>(c.lnb + cc);
  }
  const float invD = 1.f / (float)D;
  const float* s0 = c.slabs + (long long)group * nparts * 32 * D;
  for (int rr = rr0; rr < 32; rr += RPP) {       // (uniform trip count: the shuffles below see whole rows)
    const bool live = rr < nrows;
    const long long grow = (row0 + (long long)(live ? rr : 0) * rstep) * c.ld;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nparts > 1) {
      const float* sp = s0 + rr * D + cc;
      for (int z0 = 0; z0 < nparts; z0 += 8) {   // 8 partial loads in flight, then added in slot order
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = z0 + u < nparts ? *reinterpret_cast<const float4*>(sp + (long long)(z0 + u) * 32 * D) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w;
        }
      }
    } else {
      v = *reinterpret_cast<const float4*>(part + rr * SP + cc);
      for (int s = 1; s < ks; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(part + (s * 32 + rr) * SP + cc);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
    }
    const float4 r4 = *reinterpret_cast<const float4*>(c.res + grow + cc);
    v.x += r4.x + bias.x; v.y += r4.y + bias.y; v.z += r4.z + bias.z; v.w += r4.w + bias.w;
    if (live) *reinterpret_cast<float4*>(c.out + grow + cc) = v;
    if (c.lnw) {
      // LayerNorm over the row: the C4 lanes tid % C4 of one wave hold it (two-pass mean / variance like layernorm_fwd_kernel)
      float s = (v.x + v.y) + (v.z + v.w);
      for (int o = C4 >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mu = s * invD;
      const float d0 = v.x - mu, d1 = v.y - mu, d2 = v.z - mu, d3 = v.w - mu;
      float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      for (int o = C4 >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
      const float rs = rsqrtf(q * invD + 1e-5f);
      if (live)
        *reinterpret_cast<float4*>(c.ln_out + grow + cc) = make_float4(d0 * rs * g.x + be.x, d1 * rs * g.y + be.y, d2 * rs * g.z + be.z, d3 * rs * g.w + be.w);
    }
  }
}

// ------------------------------------------------------------------------------------------------ attention half of a block
struct AttnBlockArgs {
  int B, N, D, H, dh, I, NQ;          // NQ: query tiles per frame that are computed (all of them, or 1: the tile of token 0)
  const float* ln;                    // (B * N, D)  LayerNorm1 rows
  const float* wqkv; const float* wout;
  float scale;
  CombineArgs c;                      // res = x, out = xmid, LayerNorm2 -> ln2
};

template <int NKT>   // 32-key tiles: N <= 32 NKT
__global__ void __launch_bounds__(256) attn_block_kernel(const AttnBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NP = 32 * NKT;
  const int D = a.D, dh = a.dh, SD = D + 4, SH = dh + 4;
  // LDS: ln image [max(NP, 128 / (D / 32) ...)][SD] (later the to_out partial, up to 4 k-slices of 32 rows), k / v images [NP][SH],
  // q image [32][SH] (later the attention output), per-tile PV partials [NKT][32][SH], softmax statistics, the arrival flag
  const int ks_out = out_ksplit(D);
  const int ln_rows = NP > 32 * ks_out ? NP : 32 * ks_out;
  float* lns = smem;
  float* ksm = lns + ln_rows * SD;
  float* vsm = ksm + NP * SH;
  float* qsm = vsm + NP * SH;
  float* pos = qsm + 32 * SH;
  float* stm = pos + NKT * 32 * SH;          // [NKT][32] local maxima
  float* stl = stm + NKT * 32;               // [NKT][32] local sums
  int* flag = reinterpret_cast<int*>(stl + NKT * 32);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int qt = blockIdx.x % a.NQ, hd = (blockIdx.x / a.NQ) % a.H, b = blockIdx.x / (a.NQ * a.H);

  // ---- the frame's LayerNorm1 rows -> LDS (rows >= N are zero: their q / k / v are 0, the keys are masked below)
  {
    const int D4 = D >> 2;
    const float* src = a.ln + (long long)b * a.N * D;
    for (int f = tid; f < NP * D4; f += 256) {
      const int row = f / D4, c4 = (f - row * D4) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < a.N) v = *reinterpret_cast<const float4*>(src + (long long)row * D + c4);
      *reinterpret_cast<float4*>(lns + row * SD + c4) = v;
    }
  }
  __syncthreads();
  // ---- k, v of every row, q of this query tile: (2 NKT + 1) x (dh / 32) blocks of 32 x 32, contraction over D
  {
    const int ct_n = dh >> 5, nblk = (2 * NKT + 1) * ct_n;
    const float qscale = a.scale * DGVIT_LOG2E_F;
    for (int blk = wave; blk < nblk; blk += 4) {
      const int ct = blk % ct_n, rb = blk / ct_n;                  // rb: 0 .. NKT-1 k tiles, NKT .. 2 NKT - 1 v tiles, 2 NKT the q tile
      const int mat = rb < NKT ? 1 : (rb < 2 * NKT ? 2 : 0), rt = rb < NKT ? rb : (rb < 2 * NKT ? rb - NKT : qt);
      f32x16 acc;
      zero16(acc);
      mm_lds_x_wrow(acc, lns + (rt * 32 + li) * SD, a.wqkv + ((long long)mat * a.I + hd * dh + ct * 32 + li) * D, D, h);
      float* dst = (mat == 0 ? qsm : (mat == 1 ? ksm + rt * 32 * SH : vsm + rt * 32 * SH)) + ct * 32 + li;
      const float mul = mat == 0 ? qscale : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[arow(r, h) * SH] = acc[r] * mul;
    }
  }
  __syncthreads();
  // ---- scores of key tile kt = wave, transposed (S^T[key][query]: keys in the accumulator registers, the query on the lane)
  f32x16 s;
  zero16(s);
  if (wave < NKT) {
    const float* qrow = qsm + li * SH;
    const float* krow = ksm + (wave * 32 + li) * SH;
    for (int g = 0; g < (dh >> 3); ++g) {
      const float4 qf = *reinterpret_cast<const float4*>(qrow + 8 * g + 4 * h);
      const float4 kf = *reinterpret_cast<const float4*>(krow + 8 * g + 4 * h);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf.x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf.y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf.z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf.w, s, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = wave * 32 + arow(r, h) < a.N ? s[r] : -INFINITY;
      s[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (h == 0) stm[wave * 32 + li] = mx;
  }
  __syncthreads();
  if (wave < NKT) {
    float mx = stm[li];
#pragma unroll
    for (int kt = 1; kt < NKT; ++kt) mx = fmaxf(mx, stm[kt * 32 + li]);     // (every tile holds a real key: its maximum is finite)
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(s[r] - mx);
      s[r] = p;
      l += p;
    }
    l += __shfl_xor(l, 32, 64);
    if (h == 0) stl[wave * 32 + li] = l;
    // O^T[d][query] over this tile's keys: the probability registers are the B operand as they stand
    for (int dt = 0; dt < (dh >> 5); ++dt) {
      f32x16 o;
      zero16(o);
#pragma unroll
      for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vsm[(wave * 32 + arow(r, h)) * SH + dt * 32 + li], s[r], o, 0, 0, 0);
      float* orow = pos + (wave * 32 + li) * SH + dt * 32;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) *reinterpret_cast<float4*>(orow + 8 * c4 + 4 * h) = make_float4(o[4 * c4], o[4 * c4 + 1], o[4 * c4 + 2], o[4 * c4 + 3]);
    }
  }
  __syncthreads();
  // ---- attention output of the query tile: key-tile partials added in tile order, divided by the row sum -> q image
  {
    const int H4 = dh >> 2;
    for (int f = tid; f < 32 * H4; f += 256) {
      const int row = f / H4, c4 = (f - row * H4) * 4;
      float4 v = *reinterpret_cast<const float4*>(pos + row * SH + c4);
      float l = stl[row];
#pragma unroll
      for (int kt = 1; kt < NKT; ++kt) {
        const float4 t = *reinterpret_cast<const float4*>(pos + (kt * 32 + row) * SH + c4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        l += stl[kt * 32 + row];
      }
      const float inv = 1.f / l;
      *reinterpret_cast<float4*>(qsm + row * SH + c4) = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    }
  }
  __syncthreads();
  // ---- this head's share of to_out (32 x D, contraction over dh) -> LDS (over the ln image, dead by now), then the combine step
  const int ks = rows32_x_w_to_lds(lns, qsm, SH, a.wout, a.I, hd * dh, D, dh, wave, li, h);
  __syncthreads();
  const int nrows = a.N - qt * 32 < 32 ? a.N - qt * 32 : 32;
  publish_and_combine(a.c, lns, ks, D, b * a.NQ + qt, hd, a.H, (long long)b * a.N + qt * 32, 1, nrows, flag, tid);
}

// ------------------------------------------------------------------------------------------------ feed-forward half of a block
struct MlpBlockArgs {
  int tok, D, M, C;                   // token rows handled, widths, chunks (M / 128)
  long long rstep;                    // logical row r is global row r * rstep (1, or N: the token-0 rows of the pruned last block)
  const float* ln;                    // LayerNorm2 rows (row stride c.ld)
  const float* w1; const float* b1; const float* w2;
  CombineArgs c;                      // res = xmid, out = xout, the next block's LayerNorm1 -> ln1 (or none)
};

__global__ void __launch_bounds__(256) mlp_block_kernel(const MlpBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, SD = D + 4, SA = BHC + 4;
  const int ks_out = out_ksplit(D);
  float* xs = smem;                       // [32][SD]  LayerNorm2 rows
  float* as = xs + 32 * SD;               // [32][SA]  gelu(hidden chunk)
  float* part = as + 32 * SA;             // [ks][32][SD]
  int* flag = reinterpret_cast<int*>(part + ks_out * 32 * SD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int c = blockIdx.x % a.C, rt = blockIdx.x / a.C;
  const int nrows = a.tok - rt * 32 < 32 ? a.tok - rt * 32 : 32;
  {
    const int D4 = D >> 2;
    for (int f = tid; f < 32 * D4; f += 256) {
      const int row = f / D4, c4 = (f - row * D4) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nrows) v = *reinterpret_cast<const float4*>(a.ln + (long long)(rt * 32 + row) * a.rstep * a.c.ld + c4);
      *reinterpret_cast<float4*>(xs + row * SD + c4) = v;
    }
  }
  __syncthreads();
  // hidden chunk: wave w owns columns c 128 + 32 w .. + 31, contraction over D
  {
    const int jn = c * BHC + wave * 32 + li;
    const float bias = a.b1[jn];
    f32x16 acc;
    zero16(acc);
    mm_lds_x_wrow(acc, xs + li * SD, a.w1 + (long long)jn * D, D, h);
    float* dst = as + wave * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[arow(r, h) * SA] = gelu_erf(acc[r] + bias);
  }
  __syncthreads();
  const int ks = rows32_x_w_to_lds(part, as, SA, a.w2, a.M, c * BHC, D, BHC, wave, li, h);
  __syncthreads();
  publish_and_combine(a.c, part, ks, D, rt, c, a.C, (long long)rt * 32 * a.rstep, a.rstep, nrows, flag, tid);
}

size_t attn_block_lds(int NKT, int D, int dh) {
  const int NP = 32 * NKT, ks = out_ksplit(D), ln_rows = NP > 32 * ks ? NP : 32 * ks;
  return sizeof(float) * ((size_t)ln_rows * (D + 4) + (size_t)(2 * NP + 32 + NKT * 32) * (dh + 4) + 2 * NKT * 32 + 4);
}
size_t mlp_block_lds(int D) {
  const int ks = out_ksplit(D);
  return sizeof(float) * ((size_t)32 * (D + 4) + 32 * (BHC + 4) + (size_t)ks * 32 * (D + 4) + 4);
}

template <int NKT>
int launch_attn(const AttnBlockArgs& a, hipStream_t st) {
  const size_t lds = attn_block_lds(NKT, a.D, a.dh);
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_kernel<NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "attn_block_kernel: hipFuncSetAttribute failed");
    once.mark(bit);
  }
  const int slot = profile_begin(PROF_OTHER, 0.0, st);
  hipLaunchKernelGGL(attn_block_kernel<NKT>, dim3(a.B * a.H * a.NQ), dim3(256), lds, st, a);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("attn_block_kernel");
  return DGVIT_OK;
}

}  // namespace

// Shapes the two-launch blocks take: up to 128 tokens per frame (four key tiles), widths in multiples of 32, the LDS images of one
// workgroup within 160 KB (N = 65 at D = 256 is over: that shape stays on the GEMM schedule).
bool block_path_supports(int B, int N, int D, int H, int dh, int M) {
  // (D a power of two: a finished row sits on D / 4 adjacent lanes of one wave, reduced with an xor butterfly)
  if (B <= 0 || N <= 0 || N > 128 || (D != 32 && D != 64 && D != 128 && D != 256) || (dh != 32 && dh != 64) || H <= 0 || M <= 0 || M % BHC != 0) return false;
  const int NKT = (N + 31) / 32;
  return attn_block_lds(NKT, D, dh) <= 160 * 1024 && mlp_block_lds(D) <= 160 * 1024;
}

// scratch of the combine steps: partial rows (floats) and arrival counters (ints) for `B` frames
long long block_path_slab_floats(int B, int N, int D, int H, int M) {
  const long long NQ = (N + 31) / 32, T = (long long)B * N, RT = (T + 31) / 32;
  const long long attn = (long long)B * NQ * H * 32 * D, mlp = RT * (M / BHC) * 32 * D;
  return attn > mlp ? attn : mlp;
}
long long block_path_counters(int B, int N) {
  const long long NQ = (N + 31) / 32, T = (long long)B * N, RT = (T + 31) / 32;
  return B * NQ > RT ? B * NQ : RT;
}

// One transformer block.  x (T, D): the residual stream entering the block; ln1 (T, D): its LayerNorm1 rows (block 0: from
// layernorm_fwd, later blocks: written by the previous call); xmid, ln2 (T, D): scratch; xout (T, D): the block's output; lp: the
// block's eleven parameters in table order; next_ln (2 pointers or null): the NEXT block's LayerNorm1 weight / bias, applied to xout
// into ln1 (in place: ln1 is dead once the attention kernel of this block has read it).  token0_only: the block's output is read at
// token 0 of every frame only (the last block, pool = 'cls').  slabs / counters: block_path_slab_floats / block_path_counters
// (counters zero on entry, left zero).
int block_path_layer(const float* x, float* ln1, float* xmid, float* ln2, float* xout, const float* const* lp, const float* const* next_ln,
                     int token0_only, float* slabs, int* counters, int B, int N, int D, int H, int dh, int M, hipStream_t st) {
  DGVIT_CHECK_ARG(block_path_supports(B, N, D, H, dh, M), "block path: unsupported shape");
  enum { L_LN1W = 0, L_LN1B, L_QKV, L_OUTW, L_OUTB, L_LN2W, L_LN2B, L_FC1W, L_FC1B, L_FC2W, L_FC2B };
  const int NKT = (N + 31) / 32;
  AttnBlockArgs aa = {};
  aa.B = B; aa.N = N; aa.D = D; aa.H = H; aa.dh = dh; aa.I = H * dh; aa.NQ = token0_only ? 1 : NKT;
  aa.ln = ln1; aa.wqkv = lp[L_QKV]; aa.wout = lp[L_OUTW];
  aa.scale = 1.0f / sqrtf((float)dh);
  aa.c.slabs = slabs; aa.c.counters = counters; aa.c.bias = lp[L_OUTB]; aa.c.res = x; aa.c.out = xmid;
  aa.c.lnw = lp[L_LN2W]; aa.c.lnb = lp[L_LN2B]; aa.c.ln_out = ln2; aa.c.ld = D;
  int rc;
  switch (NKT) {
    case 1: rc = launch_attn<1>(aa, st); break;
    case 2: rc = launch_attn<2>(aa, st); break;
    case 3: rc = launch_attn<3>(aa, st); break;
    default: rc = launch_attn<4>(aa, st); break;
  }
  if (rc) return rc;
  MlpBlockArgs ma = {};
  ma.tok = token0_only ? B : B * N; ma.D = D; ma.M = M; ma.C = M / BHC; ma.rstep = token0_only ? N : 1;
  ma.ln = ln2; ma.w1 = lp[L_FC1W]; ma.b1 = lp[L_FC1B]; ma.w2 = lp[L_FC2W];
  ma.c.slabs = slabs; ma.c.counters = counters; ma.c.bias = lp[L_FC2B]; ma.c.res = xmid; ma.c.out = xout;
  ma.c.lnw = next_ln ? next_ln[0] : nullptr; ma.c.lnb = next_ln ? next_ln[1] : nullptr; ma.c.ln_out = ln1; ma.c.ld = D;
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "mlp_block_kernel: hipFuncSetAttribute failed");
    once.mark(bit);
  }
  const int slot = profile_begin(PROF_OTHER, 0.0, st);
  hipLaunchKernelGGL(mlp_block_kernel, dim3(((ma.tok + 31) / 32) * ma.C), dim3(256), mlp_block_lds(D), st, ma);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("mlp_block_kernel");
  return DGVIT_OK;
}
