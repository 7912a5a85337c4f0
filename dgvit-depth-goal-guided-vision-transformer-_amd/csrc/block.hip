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
//       ln      = LayerNorm1 rows of frame b                        (written by the previous kernel's combine step; block 0: formed here)
//       k, v    = ln Wk_h^T, ln Wv_h^T   for all N rows;  q = ln Wq_h^T for the 32 rows of qt      (MFMA, weights straight from L2)
//       ao      = softmax(q k^T dh^-1/2) v                          one wave per 32-key tile, scores transposed (keys in registers)
//       part    = ao Wout[:, h dh : (h+1) dh]^T                     this head's share of to_out, 32 x D
//                 (plain stores: the next launch adds the H shares, the kernel boundary is the hand-off)
//   mlp_block_kernel    one workgroup per (32-row tile rt, chunk c of 128 hidden units)         -- the "fused no-grad MLP"
//       xmid    = x + b_out + sum_h part[h]          in head order; every chunk workgroup of a row tile forms the same rows
//       ln2     = LayerNorm2(xmid)
//       a       = gelu(ln2 W1_c^T + b1_c)            32 x 128, stays in LDS: the hidden layer never exists in HBM
//       part    = a W2[:, c 128 : (c+1) 128]^T       32 x D
//       combine : the last of the M / 128 workgroups of rt sums the shares in chunk order, adds bias + residual -> xout, and
//                 normalises the rows with the NEXT block's LayerNorm1 -> ln1
//
// Redundant work instead of synchronisation: every (b, h, qt) workgroup projects K and V of the whole frame and every chunk workgroup
// sums the attention shares and normalises its rows itself (the chip is empty at these sizes; sharing would cost a hand-off); the one
// hand-off that remains, the sum over the 16 hidden chunks, is the split-K recipe of gemm.hip: write-through
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

// phase stamps (diagnostic build only, dgvit_set_block_stamps): thread 0 of workgroup 0 records the 100 MHz wall clock at the phase
// boundaries of both kernels -- [kernel][16] int64, attention at 0, MLP at 16 (tools/block_stamps.py)
#ifdef DGVIT_DIAG
#define BSTAMP(buf, i)                                                   \
  do {                                                                   \
    if ((buf) && blockIdx.x == 0 && threadIdx.x == 0) (buf)[i] = wall_clock64(); \
  } while (0)
#else
#define BSTAMP(buf, i)
#endif
#define DGVIT_LOG2E_F 1.4426950408889634f

// ---- weight fragments: one lane's share of up to eight 8-deep k-groups (64 k) of ONE weight row, straight from global memory / L2
struct WF {
  float4 v[8];
};
// NG < 0: the number of k-groups is the run-time `ng` (a guard per group: each guard is a branch that splits the block the compiler
// schedules); NG = 8: a full 64-deep chunk, straight-line code -- all eight LDS fragment reads can travel ahead of the 32 MFMAs
template <int NG>
__device__ __forceinline__ void wf_load(WF& f, const float* __restrict__ w, int ng, int h) {
#pragma unroll
  for (int u = 0; u < 8; ++u) f.v[u] = (NG > 0 || u < ng) ? *reinterpret_cast<const float4*>(w + 8 * u + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
}
// acc(32 x 32) += A W^T over those k-groups: A row of this lane in an LDS image (k contiguous)
template <int NG>
__device__ __forceinline__ void wf_mma(f32x16& acc, const float* a, const WF& f, int ng, int h) {
  if constexpr (NG > 0) {
    float4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(a + 8 * u + 4 * h);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u].x, f.v[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u].y, f.v[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u].z, f.v[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u].w, f.v[u].w, acc, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u < ng) {
        const float4 x = *reinterpret_cast<const float4*>(a + 8 * u + 4 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, f.v[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, f.v[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, f.v[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, f.v[u].w, acc, 0, 0, 0);
      }
    }
  }
}
// the first chunk of a K-deep contraction, for a caller that requests it early (K % 64 == 0: a full chunk)
__device__ __forceinline__ void wf_load_first(WF& f, const float* __restrict__ w, int K, int h) {
  if ((K & 63) == 0) wf_load<8>(f, w, 8, h);
  else wf_load<-1>(f, w, K >= 64 ? 8 : K >> 3, h);
}

// A wave's list of 32 x 32 output blocks, each  A_b (32 x K, LDS rows) x W_b^T (one weight row per lane, global):  wrow(b) / arow(b)
// give this lane's row pointers, done(b, acc) takes the finished block.  The blocks are walked as ONE stream of 64-deep chunks whose
// weight fragments are requested a chunk ahead -- across block boundaries too -- so a wave exposes one global round trip per call
// instead of one per block (at these sizes the kernels are chains of such round trips).  `f0` may arrive already loaded with the
// first chunk (requested by the caller under an earlier phase).
template <int NG, class WP, class AP, class DONE>
__device__ __forceinline__ void mm_stream_t(int nblocks, int K, WP wrow, AP arow_, DONE done, int h, WF& f0, bool preloaded) {
  const int nchunk = (K + 63) >> 6, nsteps = nblocks * nchunk;
  if (nsteps <= 0) return;
  auto ngs = [&](int c) { const int r = (K - 64 * c) >> 3; return r > 8 ? 8 : r; };
  WF f1;
  f32x16 acc;
  zero16(acc);
  auto step = [&](int s, const WF& f) {
    const int b = s / nchunk, c = s - b * nchunk;
    if (c == 0) zero16(acc);
    wf_mma<NG>(acc, arow_(b) + 64 * c, f, ngs(c), h);
    if (c == nchunk - 1) done(b, acc);
  };
  auto fetch = [&](int s, WF& f) {
    const int b = s / nchunk, c = s - b * nchunk;
    wf_load<NG>(f, wrow(b) + 64 * c, ngs(c), h);
  };
  if (!preloaded) fetch(0, f0);
  for (int s = 0; s < nsteps; s += 2) {
    if (s + 1 < nsteps) fetch(s + 1, f1);
    step(s, f0);
    if (s + 2 < nsteps) fetch(s + 2, f0);
    if (s + 1 < nsteps) step(s + 1, f1);
  }
}
template <class WP, class AP, class DONE>
__device__ __forceinline__ void mm_stream(int nblocks, int K, WP wrow, AP arow_, DONE done, int h, WF& f0, bool preloaded) {
  if ((K & 63) == 0) mm_stream_t<8>(nblocks, K, wrow, arow_, done, h, f0, preloaded);      // every width of the shipped / DGViT-small models
  else mm_stream_t<-1>(nblocks, K, wrow, arow_, done, h, f0, preloaded);
}

__host__ __device__ __forceinline__ int out_ksplit(int D) { return D >= 128 ? 1 : 128 / D; }   // D in {32, 64, 128, 256}: 4, 2, 1, 1

// part[ks][32][D + 4] (LDS) = k-slices of  A (32 x K, LDS rows a_img, stride sa)  x  W[j][koff + k]^T  for j in [0, D): the D / 32 column
// tiles are dealt over the four waves; with fewer than four tiles (D = 64, 32) the waves also split K (ks = 4 / tiles slices, summed
// by the reader in slice order).  Returns ks.  rows32_w_ptr gives the wave's first weight pointer (for an early wf_load).
__device__ __forceinline__ const float* rows32_w_ptr(const float* __restrict__ W, int ldw, int koff, int D, int K, int wave, int li) {
  const int nt = D >> 5, ks = out_ksplit(D), kslice = K / ks;
  const int t = nt >= 4 ? wave : wave % nt, kp = nt >= 4 ? 0 : wave / nt;
  return W + (long long)(t * 32 + li) * ldw + koff + kp * kslice;
}
__device__ __forceinline__ int rows32_x_w_to_lds(float* part, const float* a_img, int sa, const float* __restrict__ W, int ldw, int koff, int D,
                                                 int K, int wave, int li, int h, WF& f0, bool preloaded) {
  const int nt = D >> 5, SP = D + 4;
  const int ks = out_ksplit(D);
  const int kslice = K / ks;
  const int t0 = nt >= 4 ? wave : wave % nt, kp = nt >= 4 ? 0 : wave / nt;
  const int nblocks = nt >= 4 ? (nt - wave + 3) / 4 : 1;
  mm_stream(nblocks, kslice,
            [&](int b) { return W + (long long)((t0 + 4 * b) * 32 + li) * ldw + koff + kp * kslice; },
            [&](int b) { return a_img + li * sa + kp * kslice; },
            [&](int b, const f32x16& acc) {
              float* dst = part + (long long)kp * 32 * SP + (t0 + 4 * b) * 32 + li;
#pragma unroll
              for (int r = 0; r < 16; ++r) dst[arow(r, h) * SP] = acc[r];
            },
            h, f0, preloaded);
  return ks;
}

// sum over the W adjacent lanes that hold a row (W a power of two): compile-time offsets, so the steps are DPP / swizzle moves rather
// than ds_bpermute round trips through the LDS crossbar (a run-time offset costs ~100 cycles per step, and a kernel this short notices)
template <int W>
__device__ __forceinline__ float group_sum_w(float v) {
#pragma unroll
  for (int o = W >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float group_sum(float v, int C4) {
  switch (C4) {
    case 8: return group_sum_w<8>(v);
    case 16: return group_sum_w<16>(v);
    case 32: return group_sum_w<32>(v);
    default: return group_sum_w<64>(v);
  }
}
// LayerNorm (eps 1e-5, affine; two-pass mean / variance like layernorm_fwd_kernel) of one row piece: the D / 4 adjacent lanes
// tid % C4 of a wave hold the row (C4 a power of two)
__device__ __forceinline__ float4 ln_piece4(const float4 v, const float4 g, const float4 be, int C4, float invD) {
  const float mu = group_sum((v.x + v.y) + (v.z + v.w), C4) * invD;
  const float d0 = v.x - mu, d1 = v.y - mu, d2 = v.z - mu, d3 = v.w - mu;
  const float rs = rsqrtf(group_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3), C4) * invD + 1e-5f);
  return make_float4(d0 * rs * g.x + be.x, d1 * rs * g.y + be.y, d2 * rs * g.z + be.z, d3 * rs * g.w + be.w);
}

// ------------------------------------------------------------------------------------------------ attention half of a block
struct AttnBlockArgs {
  int B, N, D, H, dh, I, NQ;          // NQ: query tiles per frame that are computed (all of them, or 1: the tile of token 0)
  const float* rows;                  // (B * N, D): LayerNorm1 rows (lnw == null), or the residual stream itself (lnw != null: block 0,
  const float* lnw; const float* lnb; //             whose LayerNorm1 is applied here)
  // block 0 can also ASSEMBLE the token rows it normalises (four launches fewer per forward): row 0 of a frame = goal + pos[0]
  // (GoalFormer.py:160-162; rows 1 .. P come from the patch-embedding GEMM, positional embedding included), train-mode emb-dropout on all
  // of them (:163; the Philox mask of dropout_kernel, same seed -> same mask), the assembled rows stored to `xres` by the head-0
  // workgroups (the residual the MLP kernel reads), and the arrival counters of the MLP kernels zeroed by workgroup 0
  const float* goal; const float* pos0;           // (B, D), (D); goal == null: the rows arrive assembled
  float* xres;                                    // (B * N, D)
  float keep; unsigned long long seed; const unsigned long long* seed_dev;
  int* zero_counters; int n_counters;
  const float* wqkv; const float* wout;
  float scale;
  float* part;                        // (B, NQ, H, 32, D): every head's share of to_out for the 32 rows of a query tile
  long long* stamps;                  // diagnostic build: phase stamps (null in the product)
};

template <int NKT>   // 32-key tiles: N <= 32 NKT
__global__ void __launch_bounds__(256) attn_block_kernel(const AttnBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NP = 32 * NKT;
  const int D = a.D, dh = a.dh, SD = D + 4, SH = dh + 4;
  // LDS: ln image [ln_rows][SD] (later the to_out partial: up to 4 k-slices of 32 rows), k / v images [NP][SH], q image [32][SH]
  // (later the attention output), per-key-tile PV partials [NKT][32][SH], softmax statistics
  const int ks_out = out_ksplit(D);
  const int ln_rows = NP > 32 * ks_out ? NP : 32 * ks_out;
  float* lns = smem;
  float* ksm = lns + ln_rows * SD;
  float* vsm = ksm + NP * SH;
  float* qsm = vsm + NP * SH;
  float* pos = qsm + 32 * SH;
  float* stm = pos + NKT * 32 * SH;          // [NKT][32] local maxima
  float* stl = stm + NKT * 32;               // [NKT][32] local sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int qt = blockIdx.x % a.NQ, hd = (blockIdx.x / a.NQ) % a.H, b = blockIdx.x / (a.NQ * a.H);
  const int ct_n = dh >> 5, nblk = (2 * NKT + 1) * ct_n;
  // block list of the projection: rb 0 .. NKT-1 k tiles, NKT .. 2 NKT - 1 v tiles, 2 NKT the q tile; ct the 32-column tile of the head
  auto blk_mat = [&](int blk) { const int rb = blk / ct_n; return rb < NKT ? 1 : (rb < 2 * NKT ? 2 : 0); };
  auto blk_rt = [&](int blk) { const int rb = blk / ct_n; return rb < NKT ? rb : (rb < 2 * NKT ? rb - NKT : qt); };
  auto blk_w = [&](int blk) { return a.wqkv + ((long long)blk_mat(blk) * a.I + hd * dh + (blk % ct_n) * 32 + li) * D; };
  BSTAMP(a.stamps, 0);
  WF f0;
  const int nmine = wave < nblk ? (nblk - wave + 3) / 4 : 0;
  if (nmine > 0) wf_load_first(f0, blk_w(wave), D, h);      // the first weight fragment travels under the row staging

  // ---- the frame's LayerNorm1 rows -> LDS (rows >= N are zero: their q / k / v are 0, the keys are masked below)
  {
    const int D4 = D >> 2, d4s = __builtin_ctz(D4), c4 = (tid & (D4 - 1)) * 4;     // (D4 a power of two)
    const float invD = 1.f / (float)D;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), be = g;
    if (a.lnw) {
      g = *reinterpret_cast<const float4*>(a.lnw + c4);
      be = *reinterpret_cast<const float4*>(a.lnb + c4);
    }
    const float* src = a.rows + (long long)b * a.N * D;
    if (a.zero_counters && blockIdx.x == 0)
      for (int i = tid; i < a.n_counters; i += 256) a.zero_counters[i] = 0;       // (first used by the NEXT launch)
    unsigned long long seed = a.seed;
    if (a.goal && a.keep < 1.f && a.seed_dev) seed = *a.seed_dev;                 // graph-capturable form: the seed lives in device memory
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.goal) {
      const float4 gv = *reinterpret_cast<const float4*>(a.goal + (long long)b * D + c4), pv = *reinterpret_cast<const float4*>(a.pos0 + c4);
      g0 = make_float4(gv.x + pv.x, gv.y + pv.y, gv.z + pv.z, gv.w + pv.w);
    }
    // eight row pieces per thread in flight (one dependent round trip per eight, not per piece); NP * D4 is a multiple of 256, so
    // whole waves fall out of the guards together and the LayerNorm shuffles always see whole rows
    for (int f0 = tid; f0 < NP * D4; f0 += 256 * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        // (unconditional loads from a clamped row, zeroed afterwards: a load under a run-time condition is a branch around it and a
        //  full wait behind it -- the eight loads would become eight dependent round trips)
        const int f = f0 + u * 256, row = f >> d4s;
        const float4 t = *reinterpret_cast<const float4*>(src + (long long)(row < a.N ? row : (a.goal ? 1 : 0)) * D + c4);
        v[u] = row < a.N ? t : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int f = f0 + u * 256, row = f >> d4s;
        if (f < NP * D4) {
          if (a.goal) {
            if (row == 0) v[u] = g0;
            if (a.keep < 1.f && row < a.N) v[u] = dropout4(v[u], (((long long)b * a.N + row) * D + c4) >> 2, seed, a.keep);
            // the assembled rows of this query tile go back to memory once per frame (head 0): the MLP kernel's residual
            if (hd == 0 && row < a.N && (row >> 5) == qt) *reinterpret_cast<float4*>(a.xres + ((long long)b * a.N + row) * D + c4) = v[u];
          }
          if (a.lnw) {
            const float4 y = ln_piece4(v[u], g, be, D4, invD);
            if (row < a.N) v[u] = y;
          }
          *reinterpret_cast<float4*>(lns + row * SD + c4) = v[u];
        }
      }
    }
  }
  __syncthreads();
  BSTAMP(a.stamps, 1);
  // ---- k, v of every row, q of this query tile: (2 NKT + 1) x (dh / 32) blocks of 32 x 32, contraction over D
  {
    const float qscale = a.scale * DGVIT_LOG2E_F;
    mm_stream(nmine, D,
              [&](int i) { return blk_w(wave + 4 * i); },
              [&](int i) { return lns + (blk_rt(wave + 4 * i) * 32 + li) * SD; },
              [&](int i, const f32x16& acc) {
                const int blk = wave + 4 * i, mat = blk_mat(blk), rt = blk_rt(blk);
                float* dst = (mat == 0 ? qsm : (mat == 1 ? ksm + rt * 32 * SH : vsm + rt * 32 * SH)) + (blk % ct_n) * 32 + li;
                const float mul = mat == 0 ? qscale : 1.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[arow(r, h) * SH] = acc[r] * mul;
              },
              h, f0, true);
  }
  // the first to_out weight fragment of this wave travels under the softmax
  WF fo;
  wf_load_first(fo, rows32_w_ptr(a.wout, a.I, hd * dh, D, dh, wave, li), dh / ks_out, h);
  __syncthreads();
  BSTAMP(a.stamps, 2);
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
  BSTAMP(a.stamps, 3);
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
  BSTAMP(a.stamps, 4);
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
  BSTAMP(a.stamps, 5);
  // ---- this head's share of to_out (32 x D, contraction over dh) -> LDS (over the ln image, dead by now) -> global, k-slices summed.
  // Plain stores: the consumer is the NEXT launch (mlp_block_kernel adds the H shares in head order), the kernel boundary is the hand-off.
  const int ks = rows32_x_w_to_lds(lns, qsm, SH, a.wout, a.I, hd * dh, D, dh, wave, li, h, fo, true);
  __syncthreads();
  BSTAMP(a.stamps, 6);
  {
    const int C4 = D >> 2, RPP = 256 / C4, cc = (tid % C4) * 4;
    float* dst = a.part + (((long long)b * a.NQ + qt) * a.H + hd) * 32 * D;
    for (int rr = tid / C4; rr < 32; rr += RPP) {
      float4 v = *reinterpret_cast<const float4*>(lns + rr * SD + cc);
      for (int z = 1; z < ks; ++z) {
        const float4 t = *reinterpret_cast<const float4*>(lns + (z * 32 + rr) * SD + cc);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      *reinterpret_cast<float4*>(dst + rr * D + cc) = v;
    }
  }
  BSTAMP(a.stamps, 7);
}

// ------------------------------------------------------------------------------------------------ feed-forward half of a block
struct MlpBlockArgs {
  int tok, N, D, H, M, C, NQ;         // token rows handled; tokens per frame; widths; chunks (M / 128); query tiles the attention kernel ran
  long long rstep;                    // logical row r is global token row r * rstep (1, or N: the token-0 rows of the pruned last block)
  const float* x;                     // (T, D) residual stream entering the block
  const float* apart;                 // (B, NQ, H, 32, D) the attention kernel's shares of to_out
  const float* bout;                  // (D) to_out bias
  const float* ln2w; const float* ln2b;
  const float* w1; const float* b1; const float* w2; const float* b2;
  float* slabs; int* counters;        // [row tile][chunk][32][D] partial rows; one arrival counter per row tile (zero on entry, left zero)
  float* out;                         // (T, D) the block's output rows
  const float* lnw; const float* lnb; float* ln_out;   // the NEXT block's LayerNorm1 applied to them (or null)
  const float* rms_g; float* feat;    // last block, pool = 'cls': feat[r] = F.normalize(row r) * sqrt(D) * g (GoalFormer.py:120-122,170), r = the tile's logical rows
  int sc1_reads;                      // the last arriver reads the other workgroups' partials with sc1 loads instead of acquiring
  long long* stamps;                  // diagnostic build: phase stamps (null in the product)
};

__global__ void __launch_bounds__(256) mlp_block_kernel(const MlpBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, SD = D + 4, SA = BHC + 4;
  const int ks_out = out_ksplit(D);
  float* xm = smem;                       // [32][SD]  xmid rows (kept for the residual of the output)
  float* xs = xm + 32 * SD;               // [32][SD]  LayerNorm2 rows
  float* as = xs + 32 * SD;               // [32][SA]  gelu(hidden chunk)
  float* part = as + 32 * SA;             // [ks][32][SD]
  int* flag = reinterpret_cast<int*>(part + ks_out * 32 * SD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int c = blockIdx.x % a.C, rt = blockIdx.x / a.C;
  const int nrows = a.tok - rt * 32 < 32 ? a.tok - rt * 32 : 32;
  const int C4 = D >> 2, RPP = 256 / C4, cc = (tid & (C4 - 1)) * 4, rr0 = tid >> __builtin_ctz(C4);      // (C4 a power of two)
  const float invD = 1.f / (float)D;
  BSTAMP(a.stamps, 16);
  // weight fragments first: W1 rows of this wave's 32 hidden columns, and the first W2 fragment of the output stage
  const int jn = c * BHC + wave * 32 + li;
  WF f1, f2;
  wf_load_first(f1, a.w1 + (long long)jn * D, D, h);
  wf_load_first(f2, rows32_w_ptr(a.w2, a.M, c * BHC, D, BHC, wave, li), BHC / ks_out, h);
  const float bias1 = a.b1[jn];
  // ---- xmid = x + b_out + sum over heads of the attention kernel's shares (head order), then LayerNorm2; every chunk workgroup of
  // the row tile forms the same rows in the same order (redundant work instead of a hand-off)
  {
    const float4 bo = *reinterpret_cast<const float4*>(a.bout + cc);
    const float4 g = *reinterpret_cast<const float4*>(a.ln2w + cc), be = *reinterpret_cast<const float4*>(a.ln2b + cc);
    // two rows per thread in flight at a time: 2 x (1 + up to 8 head shares) 16-byte loads, one dependent round trip per pair
    for (int rb = rr0; rb < 32; rb += 2 * RPP) {       // (uniform trip count; RPP = 32 at D = 32: one row per thread, the pair's second is idle)
      float4 v[2], t[2][8];
      bool live[2];
      const float* ap[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int rr = rb + e * RPP;
        live[e] = rr < 32 && rr < nrows;
        const int grow = (rt * 32 + (live[e] ? rr : 0)) * (int)a.rstep;       // global token row (< 2^31 / D by the launch's size bound)
        const int fb = a.rstep == 1 ? grow / a.N : grow / (int)a.rstep, n = a.rstep == 1 ? grow - fb * a.N : 0;   // (rstep == N: token 0 of frame fb)
        ap[e] = a.apart + (long long)(((fb * a.NQ + (n >> 5)) * a.H) * 32 + (n & 31)) * D + cc;
        v[e] = *reinterpret_cast<const float4*>(a.x + (long long)grow * D + cc);
        // (unconditional loads, clamped to the last head and masked when added: a load under a run-time condition would be a branch
        //  and a full wait per load -- eight dependent round trips instead of one)
#pragma unroll
        for (int u = 0; u < 8; ++u) t[e][u] = *reinterpret_cast<const float4*>(ap[e] + (long long)(u < a.H ? u : a.H - 1) * 32 * D);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int rr = rb + e * RPP;
        if (rr >= 32) continue;                        // (only at D = 32, and then for every thread alike: the shuffles below stay whole-wave)
        v[e].x += bo.x; v[e].y += bo.y; v[e].z += bo.z; v[e].w += bo.w;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float m = u < a.H ? 1.f : 0.f;      // (x + 0 * finite: the same bits as not adding)
          v[e].x += m * t[e][u].x; v[e].y += m * t[e][u].y; v[e].z += m * t[e][u].z; v[e].w += m * t[e][u].w;
        }
        for (int u0 = 8; u0 < a.H; u0 += 8) {      // more than eight heads: the rest, eight loads at a time
#pragma unroll
          for (int u = 0; u < 8; ++u) t[e][u] = *reinterpret_cast<const float4*>(ap[e] + (long long)(u0 + u < a.H ? u0 + u : a.H - 1) * 32 * D);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float m = u0 + u < a.H ? 1.f : 0.f;
            v[e].x += m * t[e][u].x; v[e].y += m * t[e][u].y; v[e].z += m * t[e][u].z; v[e].w += m * t[e][u].w;
          }
        }
        if (!live[e]) v[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xm + rr * SD + cc) = v[e];
        const float4 y = ln_piece4(v[e], g, be, C4, invD);
        *reinterpret_cast<float4*>(xs + rr * SD + cc) = live[e] ? y : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  __syncthreads();
  BSTAMP(a.stamps, 17);
  // ---- hidden chunk: wave w owns columns c 128 + 32 w .. + 31, contraction over D; gelu -> LDS (the hidden layer never reaches HBM)
  mm_stream(1, D,
            [&](int) { return a.w1 + (long long)jn * D; },
            [&](int) { return xs + li * SD; },
            [&](int, const f32x16& acc) {
              float* dst = as + wave * 32 + li;
#pragma unroll
              for (int r = 0; r < 16; ++r) dst[arow(r, h) * SA] = gelu_erf(acc[r] + bias1);
            },
            h, f1, true);
  __syncthreads();
  BSTAMP(a.stamps, 18);
  const int ks = rows32_x_w_to_lds(part, as, SA, a.w2, a.M, c * BHC, D, BHC, wave, li, h, f2, true);
  __syncthreads();
  BSTAMP(a.stamps, 19);
  // ---- publish this chunk's 32 x D share; the last chunk workgroup of the row tile to arrive sums all of them in chunk order
  // (gemm.hip's split-K hand-off: write-through stores, every wave drains, barrier, one relaxed agent-scope ticket)
  float* slab = a.slabs + ((long long)rt * a.C + c) * 32 * D;
  if (a.C > 1) {
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, 32 * D * 4, 0x00020000);
    for (int rr = rr0; rr < 32; rr += RPP) {
      float4 v = *reinterpret_cast<const float4*>(part + rr * SD + cc);
      for (int z = 1; z < ks; ++z) {
        const float4 t = *reinterpret_cast<const float4*>(part + (z * 32 + rr) * SD + cc);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srs, (unsigned)((rr * D + cc) * 4), 0, 16);   // sc1: write-through
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    BSTAMP(a.stamps, 20);
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(a.counters + rt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == a.C - 1;
      if (last) {
        if (!a.sc1_reads) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_store(a.counters + rt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
      }
      *flag = last;
    }
    __syncthreads();
    BSTAMP(a.stamps, 21);
    if (!*flag) return;
#ifdef DGVIT_DIAG
    if (a.stamps && rt == 0 && tid == 0) a.stamps[23] = wall_clock64();      // the last arriver of row tile 0: combine starts ...
#endif
  }
  {
    const float4 b2 = *reinterpret_cast<const float4*>(a.b2 + cc);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), be = g;
    if (a.lnw) {
      g = *reinterpret_cast<const float4*>(a.lnw + cc);
      be = *reinterpret_cast<const float4*>(a.lnb + cc);
    }
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(a.slabs + (long long)rt * a.C * 32 * D, 0, a.C * 32 * D * 4, 0x00020000);
    // Sixteen chunk shares of TWO rows per thread in flight at once (32 x 16 bytes): the shares come from memory (write-through
    // stores, read past the caches), and every dependent batch is a full round trip on the critical path of the row tile.
    for (int rb = rr0; rb < 32; rb += 2 * RPP) {       // (uniform trip count: the LayerNorm shuffles see whole rows)
      float4 v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) v[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.C > 1) {
        for (int z0 = 0; z0 < a.C; z0 += 16) {
          float4 t[2][16];
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int rr = rb + e * RPP;
              const unsigned off = (z0 + u < a.C && rr < 32) ? (unsigned)((((z0 + u) * 32 + rr) * D + cc) * 4) : 0x80000000u;     // out of range reads 0
              // sc1 loads (served past this CU's L1 and this XCD's L2) where the launch asked for them, plain loads behind the acquire
              t[e][u] = a.sc1_reads ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrs, off, 0, 16))
                                    : __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrs, off, 0, 0));
            }
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int u = 0; u < 16; ++u) {       // added in chunk order
              v[e].x += t[e][u].x; v[e].y += t[e][u].y; v[e].z += t[e][u].z; v[e].w += t[e][u].w;
            }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int rr = rb + e * RPP;
          if (rr < 32) {
            v[e] = *reinterpret_cast<const float4*>(part + rr * SD + cc);
            for (int z = 1; z < ks; ++z) {
              const float4 t = *reinterpret_cast<const float4*>(part + (z * 32 + rr) * SD + cc);
              v[e].x += t.x; v[e].y += t.y; v[e].z += t.z; v[e].w += t.w;
            }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int rr = rb + e * RPP;
        if (rr >= 32) continue;                        // (only at D = 32, and then for every thread alike)
        const bool live = rr < nrows;
        const float4 r4 = *reinterpret_cast<const float4*>(xm + rr * SD + cc);
        v[e].x += r4.x + b2.x; v[e].y += r4.y + b2.y; v[e].z += r4.z + b2.z; v[e].w += r4.w + b2.w;
        const long long grow = (long long)(rt * 32 + (live ? rr : 0)) * a.rstep * D;
        if (live) *reinterpret_cast<float4*>(a.out + grow + cc) = v[e];
        if (a.lnw) {
          const float4 y = ln_piece4(v[e], g, be, C4, invD);
          if (live) *reinterpret_cast<float4*>(a.ln_out + grow + cc) = y;
        }
        if (a.rms_g) {       // the final RMSNorm of the pooled token (rmsnorm_fwd_kernel's arithmetic: x / max(|x|, 1e-12) * sqrt(D) * g)
          const float nrm = fmaxf(sqrtf(group_sum((v[e].x * v[e].x + v[e].y * v[e].y) + (v[e].z * v[e].z + v[e].w * v[e].w), C4)), 1e-12f);
          const float sc = sqrtf((float)D);
          const float4 gg = *reinterpret_cast<const float4*>(a.rms_g + cc);
          if (live)
            *reinterpret_cast<float4*>(a.feat + (long long)(rt * 32 + rr) * D + cc) =
                make_float4(v[e].x / nrm * sc * gg.x, v[e].y / nrm * sc * gg.y, v[e].z / nrm * sc * gg.z, v[e].w / nrm * sc * gg.w);
        }
      }
    }
  }
  BSTAMP(a.stamps, 22);
#ifdef DGVIT_DIAG
  if (a.stamps && rt == 0 && tid == 0) a.stamps[24] = wall_clock64();        // ... and ends
#endif
}

size_t attn_block_lds(int NKT, int D, int dh) {
  const int NP = 32 * NKT, ks = out_ksplit(D), ln_rows = NP > 32 * ks ? NP : 32 * ks;
  return sizeof(float) * ((size_t)ln_rows * (D + 4) + (size_t)(2 * NP + 32 + NKT * 32) * (dh + 4) + 2 * NKT * 32);
}
size_t mlp_block_lds(int D) {
  const int ks = out_ksplit(D);
  return sizeof(float) * ((size_t)64 * (D + 4) + 32 * (BHC + 4) + (size_t)ks * 32 * (D + 4) + 4);
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

// scratch: the attention kernel's per-head shares, then the MLP kernel's per-chunk shares (floats); one arrival counter per 32-row tile
long long block_path_slab_floats(int B, int N, int D, int H, int M) {
  const long long NQ = (N + 31) / 32, T = (long long)B * N, RT = (T + 31) / 32;
  return (long long)B * NQ * H * 32 * D + RT * (M / BHC) * 32 * D;
}
long long block_path_counters(int B, int N) { return ((long long)B * N + 31) / 32; }

// One transformer block.  x (T, D): the residual stream entering the block; ln1 (T, D): its LayerNorm1 rows (written by the previous
// call), or null for the first block, whose LayerNorm1 then runs inside the attention kernel; xout (T, D): the block's output;
// lp: the block's eleven parameters in table order; next_ln (2 pointers or null): the NEXT block's LayerNorm1 weight / bias, applied
// to xout into ln1_out.  token0_only: the block's output is read at token 0 of every frame only (the last block, pool = 'cls').
// slabs / counters: block_path_slab_floats / block_path_counters.  first (block 0 only, may be null): the attention kernel assembles
// the token rows itself -- x holds the patch rows (+ positional embedding) from the patch-embedding GEMM, row 0 of every frame is
// goal + pos0, emb-dropout (keep < 1) is applied to all of them, the result is stored to first->xres (which becomes the block's
// residual stream) and the counters are zeroed by the kernel; without it the counters must be zero on entry.  They are left zero.
// rms_g / feat (last block, token0_only): the final RMSNorm of the pooled rows runs in the MLP kernel's combine step.
int block_path_layer(const float* x, const float* ln1, float* xout, float* ln1_out, const float* const* lp, const float* const* next_ln,
                     int token0_only, float* slabs, int* counters, const BlockFirst* first, const float* rms_g, float* feat, int B, int N, int D,
                     int H, int dh, int M, hipStream_t st) {
  DGVIT_CHECK_ARG(block_path_supports(B, N, D, H, dh, M), "block path: unsupported shape");
  enum { L_LN1W = 0, L_LN1B, L_QKV, L_OUTW, L_OUTB, L_LN2W, L_LN2B, L_FC1W, L_FC1B, L_FC2W, L_FC2B };
  const int NKT = (N + 31) / 32;
  AttnBlockArgs aa = {};
  aa.B = B; aa.N = N; aa.D = D; aa.H = H; aa.dh = dh; aa.I = H * dh; aa.NQ = token0_only ? 1 : NKT;
  aa.rows = ln1 ? ln1 : x;
  aa.lnw = ln1 ? nullptr : lp[L_LN1W]; aa.lnb = ln1 ? nullptr : lp[L_LN1B];
  if (first) {
    DGVIT_CHECK_ARG(!ln1 && first->goal && first->pos0 && first->xres && first->keep > 0.f && first->keep <= 1.f, "block path: bad first-block arguments");
    aa.goal = first->goal; aa.pos0 = first->pos0; aa.xres = first->xres; aa.keep = first->keep; aa.seed = first->seed; aa.seed_dev = first->seed_dev;
    aa.zero_counters = counters; aa.n_counters = (int)block_path_counters(B, N);
    x = first->xres;       // the MLP kernel's residual: the assembled rows
  }
  aa.wqkv = lp[L_QKV]; aa.wout = lp[L_OUTW];
  aa.scale = 1.0f / sqrtf((float)dh);
  aa.part = slabs;
#ifdef DGVIT_DIAG
  aa.stamps = g_block_stamp_now ? g_block_stamps : nullptr;
#endif
  int rc;
  switch (NKT) {
    case 1: rc = launch_attn<1>(aa, st); break;
    case 2: rc = launch_attn<2>(aa, st); break;
    case 3: rc = launch_attn<3>(aa, st); break;
    default: rc = launch_attn<4>(aa, st); break;
  }
  if (rc) return rc;
  MlpBlockArgs ma = {};
  ma.tok = token0_only ? B : B * N; ma.N = N; ma.D = D; ma.H = H; ma.M = M; ma.C = M / BHC; ma.NQ = aa.NQ; ma.rstep = token0_only ? N : 1;
  ma.x = x; ma.apart = slabs; ma.bout = lp[L_OUTB]; ma.ln2w = lp[L_LN2W]; ma.ln2b = lp[L_LN2B];
  ma.w1 = lp[L_FC1W]; ma.b1 = lp[L_FC1B]; ma.w2 = lp[L_FC2W]; ma.b2 = lp[L_FC2B];
  ma.slabs = slabs + (long long)B * NKT * H * 32 * D; ma.counters = counters;
#ifdef DGVIT_DIAG
  ma.stamps = g_block_stamp_now ? g_block_stamps : nullptr;
#endif
  ma.out = xout; ma.lnw = next_ln ? next_ln[0] : nullptr; ma.lnb = next_ln ? next_ln[1] : nullptr; ma.ln_out = ln1_out;
  if (rms_g && feat && token0_only) { ma.rms_g = rms_g; ma.feat = feat; }
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "mlp_block_kernel: hipFuncSetAttribute failed");
    once.mark(bit);
  }
  const int grid = ((ma.tok + 31) / 32) * ma.C;
  // Up to one workgroup per CU (a dynamic-LDS request above half the CU's 160 KB keeps a second one out): the last arriver reads the
  // other workgroups' write-through partials with sc1 loads instead of an agent-scope acquire -- the form MI355X_MICROARCH.md measures
  // for one workgroup per CU (one lane's ticket behind every wave's drain and a barrier, the last arriver told by the returned value,
  // its other waves behind a barrier) -- and saves the ~1.7 us fence on the critical path of every row tile.  Larger grids, where
  // several workgroups share a CU, keep the acquire and plain loads.
  size_t lds = mlp_block_lds(D);
  ma.sc1_reads = grid <= 256 ? 1 : 0;
  if (ma.sc1_reads && lds < 82 * 1024) lds = 82 * 1024;
  const int slot = profile_begin(PROF_OTHER, 0.0, st);
  hipLaunchKernelGGL(mlp_block_kernel, dim3(grid), dim3(256), lds, st, ma);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("mlp_block_kernel");
  return DGVIT_OK;
}
