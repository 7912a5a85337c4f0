// Fused multi-head self-attention, bf16 storage / fp32 softmax and accumulation on v_mfma_f32_32x32x16_bf16
// (Attention.forward, GoalFormer.py:73-81, in the bf16 configuration).  Same plan as attention.hip:
// one workgroup per (frame, head), K and V of the head in LDS, one wave per 32-query tile, scores computed
// transposed (S^T[key][query] = K Q^T) so that a lane owns one query: the softmax is register-local plus one
// cross-half exchange and the probability tile, converted to bf16 pairwise, IS the B operand of O^T = V^T P^T
// (registers 8s..8s+7 form k-step s; element j of lane half h is key 16 s + 8 (j >> 2) + 4 h + (j & 3)).
//
// LDS images (64 features = 128-byte rows per token):
//   row image for ds_read_b128 fragments (K in forward; K, V, Q, dO rows in backward): 16-byte chunks XOR-swizzled by
//      ((token >> 1) & 7) -> conflict-free
//   forward V: row image read TRANSPOSED with ds_read_b64_tr_b16 (tr_frag): the contraction index of P V is the key, so a
//      lane needs 4 + 4 consecutive keys of its own feature -- the hardware transposes 4 x 16 blocks on the fly
//   backward K / Q / dO transposed images [64][NP + 8] built at staging time (stage_transposed): inside each 16-token group
//      tokens are stored at position 8*((k>>2)&1) + 4*(k>>3) + (k&3) so that the eight tokens a lane needs for one k-step are
//      16 contiguous bytes; row stride (2 NP + 16) bytes = 16 x odd: the 16 lanes of a ds_read_b128 group hit 16 distinct
//      bank slots.  (Both forms measured equal in the forward kernel: it is bound by staging latency and softmax VALU.)
#include "bf16.h"
#include "kernels.h"

namespace {

#define DGVIT_LOG2E 1.4426950408889634f
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ int vt_pos(int key) {
  const int w = key & 15;
  return (key & ~15) | (((w >> 2) & 1) << 3) | ((w >> 3) << 2) | (w & 3);
}

// transposed fragment of a row-major [token][64] image (128-byte rows, chunk c of row r stored at c ^ (((r >> 1) & 1) << 2)):
// A[row = feature 32 dt + (lane & 31)][k = the 8 tokens 16 s + 8 (j >> 2) + 4 h + (j & 3)] of token tile `t0` -- the order
// in which a 32x32 accumulator's registers 8s .. 8s+7 present their rows (see the header).  Two ds_read_b64_tr_b16: per
// 16-lane group the hardware reads 4 tokens x 16 features and hands lane i feature i; the swizzle puts the four token rows
// on four different 64-byte bank groups.  EXEC must be all ones (uniform control flow only around this).
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* img, int t0, int dt, int s, int lane) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const int w = lane & 15, q = w >> 2, p = w & 3, cb = (lane >> 4) & 1, h = lane >> 5;
  const int c = ((dt ^ ((q >> 1) & 1)) << 2) | (2 * cb + (p >> 1));
  const unsigned char* a = img + (t0 + 16 * s + 4 * h + q) * 128 + c * 16 + (p & 1) * 8;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 8 * 128));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// rows [0, N) of a 64-wide per-head column block -> that image (rows >= N zero)
template <int NTHR>
__device__ __forceinline__ void stage_rows_tr(unsigned char* img, const bf16_t* src, long long ld, int N, int NP, int tid) {
  for (int f = tid; f < NP * 8; f += NTHR) {
    const int row = f >> 3, pc = f & 7, c = pc ^ (((row >> 1) & 1) << 2);
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (row < N) v = *reinterpret_cast<const u32x4_t*>(src + row * ld + c * 8);
    *reinterpret_cast<u32x4_t*>(img + row * 128 + pc * 16) = v;
  }
}


// One key tile of the online softmax on a transposed score tile (a lane owns one query: 16 of its 32 keys in s0, the other 16 in the
// lane 32 away).  In: raw scores q.k; out: s0 = un-normalised probabilities exp2(score * sc - m), l and o brought to the running
// maximum m.  The kernel is bound by exactly this VALU work (MFMA-busy 0.15), so: the scale is folded into the exponent's fma
// (max of the raw scores, scaled once per row); keys >= N are masked in the last tile only; and the accumulator rescale is DEFERRED
// (cdna_hip_programming.md T13): m only moves when some query's tile maximum exceeds it by more than 2^DEFER in probability, so
// after the first tile the 32-register multiply of o almost never runs.  Probabilities then reach 2^DEFER instead of 1 -- the same
// relative precision in bf16, sums in fp32; the normalisation by l at the end is exact either way.
#define DGVIT_ATTN_DEFER 4.0f
template <int NT>
__device__ __forceinline__ void softmax_step(f32x16 (&s0)[NT], float& m, float& l, f32x16 (&o)[2], float sc, int kt, int nkt, int N, int h) {
  if (kt + NT == nkt && (N & 31)) {   // uniform: only the last tile has keys past N
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((nkt - 1) * 32 + acc_row(r, h) >= N) s0[NT - 1][r] = -INFINITY;
  }
  float mt = s0[0][0];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s0[t][r]);
  mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * sc;         // (sc > 0; every tile holds at least one real key: finite)
  float alpha = 1.f;
  if (!__all(mt - m <= DGVIT_ATTN_DEFER)) {            // wave-uniform; always taken in the first tile (m = -inf)
    const float mn = fmaxf(m, mt);
    alpha = __builtin_amdgcn_exp2f(m - mn);            // first tile: exp2(-inf) = 0 (l and o are 0 there)
    m = mn;
    if (kt > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] *= alpha;
        o[1][r] *= alpha;
      }
    }
  }
  float ts = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = __builtin_amdgcn_exp2f(fmaf(s0[t][r], sc, -m));
      s0[t][r] = pr;
      ts += pr;
    }
  ts += __shfl_xor(ts, 32, 64);
  l = l * alpha + ts;
}

// NT (1 or 2) key tiles of one query tile: S^T = K Q^T (independent accumulator chains), the softmax step over all of them, O^T += V^T P^T
template <int NT>
__device__ __forceinline__ void attn_key_tiles(const unsigned char* Ks, const unsigned char* Vs, const bf16x8 (&qf)[4], float& m, float& l, f32x16 (&o)[2],
                                               float sc, int kt, int nkt, int N, int li, int h, int lane, unsigned fsw) {
  f32x16 s0[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[t][r] = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ks + ((kt + t) * 32 + li) * 128 + (((2 * s + h) ^ fsw) * 16));
      s0[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], s0[t], 0, 0, 0);
    }
  softmax_step<NT>(s0, m, l, o, sc, kt, nkt, N, h);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bf16x8 pf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[s][j] = (__bf16)s0[t][8 * s + j];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vs, (kt + t) * 32, dt, s, lane), pf[s], o[dt], 0, 0, 0);
  }
}

template <int NW>
__global__ void __launch_bounds__(64 * NW) attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                float* __restrict__ lse, int N, int H, float scale, int nq) {
  constexpr int DH = 64, NTHR = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nkt = (N + 31) / 32, NP = nkt * 32;
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + NP * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int I = H * DH;
  const long long ld = 3ll * I;
  const bf16_t* base = qkv + (long long)b * N * ld + hd * DH;
  const float sc = scale * DGVIT_LOG2E;

  // ---- stage K (row image for ds_read_b128) and V (row image for transposed reads), zero padding keys >= N ----
  // all of a thread's 16-byte loads are requested before the first LDS write (a load-store-load-store loop would serialise
  // one memory round trip per chunk)
  {
    constexpr int CH = 4;   // chunks per thread, image and trip: NP * 8 <= CH * NTHR for every supported N when NTHR = 512
    for (int f0 = tid; f0 < NP * 8; f0 += CH * NTHR) {
      u32x4_t kv[CH], vv[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int f = f0 + j * NTHR, row = f >> 3, pc = f & 7;
        const bool ok = f < NP * 8 && row < N;
        const int rr = ok ? row : 0;
        kv[j] = *reinterpret_cast<const u32x4_t*>(base + I + rr * ld + (pc ^ ((row >> 1) & 7)) * 8);
        vv[j] = *reinterpret_cast<const u32x4_t*>(base + 2 * I + rr * ld + (pc ^ (((row >> 1) & 1) << 2)) * 8);
        if (!ok) {
          kv[j] = u32x4_t{0u, 0u, 0u, 0u};
          vv[j] = u32x4_t{0u, 0u, 0u, 0u};
        }
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int f = f0 + j * NTHR, row = f >> 3, pc = f & 7;
        if (f < NP * 8) {
          *reinterpret_cast<u32x4_t*>(Ks + row * 128 + pc * 16) = kv[j];
          *reinterpret_cast<u32x4_t*>(Vs + row * 128 + pc * 16) = vv[j];
        }
      }
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
    int kt = 0;
    for (; kt + 2 <= nkt; kt += 2) attn_key_tiles<2>(Ks, Vs, qf, m, l, o, sc, kt, nkt, N, li, h, lane, fsw);   // two key tiles at a time: twice
    if (kt < nkt) attn_key_tiles<1>(Ks, Vs, qf, m, l, o, sc, kt, nkt, N, li, h, lane, fsw);                   // the independent work per dependent chain
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

// ------------------------------------------------------------------------------------------------ forward, persistent form
// The per-(frame, head) kernel above is latency-bound: a workgroup stages 50 KB, waits, computes, stores, and only two of them
// fit a CU (175 us per launch at BASELINE config 5, 3 TB/s -- its HBM time is ~100 us).  Here one workgroup per CU walks items
// (frame, head) = id, id + grid, ... and the K / V / Q row images of item i + 1 are brought in by LDS-DMA (`buffer_load ... lds`,
// 8 rows x 128 bytes per instruction: whole lines, no registers) WHILE item i is computed: two K and two V buffers of 256 rows
// (32 KB each) and one Q buffer, all 160 KB of LDS.  Per item two barriers: after every wave has its Q fragments (the Q buffer
// is free for the next item's DMA), and after the compute (the next item has landed -- the loader wave waited for it -- and this
// item's K / V buffers are free; the compute waves' output stores simply stay in flight).
// Rows >= N read zero through the buffer range check (V must be finite there: its probabilities are exact zeros).
typedef __attribute__((address_space(3))) void* attn_lds_ptr_t;
template <bool HAS_LSE>
__global__ void __launch_bounds__(512) attn_fwd_bf16_stream_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse,
                                                                   int N, int H, float scale, int nq, int nitems) {
  constexpr int DH = 64, IMG = 256 * 128;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int nkt = (N + 31) / 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int I = H * DH;
  const unsigned ldb = 3u * (unsigned)I * 2u;          // bytes between token rows of qkv
  const float sc = scale * DGVIT_LOG2E;
  unsigned char* Qs = smem + 4 * IMG;
  // DMA lane geometry: instruction j (0..3) of a wave fills rows 64 j + 8 wave + (lane >> 3); physical chunk lane & 7 holds logical
  // chunk (lane & 7) ^ f(row): f = (row >> 1) & 7 for the K / Q images, ((row >> 1) & 1) << 2 for V.  (row >> 1) & 7 = (4 (wave & 1)
  // + (lane >> 4)) & 7 and (row >> 1) & 1 = (lane >> 4) & 1 for every j.
  // Wave 7 never has a query tile (N <= 224: at most seven), so it is the LOADER: it issues all 96 DMA instructions of the next item
  // (12 shares x 8), the compute waves none -- an LDS-DMA instruction blocks its wave for ~100 cycles, 12 of them in front of every
  // item's first MFMA was a tenth of the item.
  const unsigned cv = (unsigned)((lane & 7) ^ (((lane >> 4) & 1) << 2)), dstep = 64u * ldb;
  auto issue_item = [&](int item, int buf, int sw) {
    const int drow = 8 * sw + (lane >> 3);
    const unsigned ck = (unsigned)((lane & 7) ^ ((4 * (sw & 1) + (lane >> 4)) & 7));
    const unsigned offk = (unsigned)drow * ldb + ck * 16u, offv = (unsigned)drow * ldb + cv * 16u;
    // (past the last item every lane is out of range: zeros into a free buffer, the same instruction count)
    const bool live = item < nitems;
    const int b = live ? item / H : 0, hd = live ? item % H : 0;
    const bf16_t* base = qkv + (long long)b * N * (3ll * I) + hd * DH;
    const int bytes = live ? (int)((unsigned)(N - 1) * ldb + 128u) : 0;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base + I), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base + 2 * I), 0, bytes, 0x00020000);
    unsigned char* kd = smem + buf * IMG + sw * 1024, *vd = smem + (2 + buf) * IMG + sw * 1024, *qd = Qs + sw * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (attn_lds_ptr_t)(kd + j * 8192), 16, offk + j * dstep, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (attn_lds_ptr_t)(vd + j * 8192), 16, offv + j * dstep, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (attn_lds_ptr_t)(qd + j * 8192), 16, offk + j * dstep, 0, 0, 0);
  };
  const unsigned fsw = (unsigned)((li >> 1) & 7);
  const int nqt = (nq + 31) / 32;
  issue_item(blockIdx.x, 0, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x, buf ^= 1) {
    const unsigned char* Ks = smem + buf * IMG, *Vs = smem + (2 + buf) * IMG;
    const int b = item / H, hd = item % H;
    // this wave's query fragments (waves beyond the last query tile read tile 0: harmless)
    const int qt = wave < nqt ? wave : 0;
    bf16x8 qf[4];
    {
      const unsigned char* qrow = Qs + (qt * 32 + li) * 128;
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + (((2 * s + h) ^ fsw) * 16));
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();           // every wave holds its Q fragments: the Q buffer may take the next item
    if (wave == 7) {
      for (int sw = 0; sw < 8; ++sw) issue_item(item + (int)gridDim.x, buf ^ 1, sw);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // output window of this item: rows past nq are dropped by the range check, every store below is issued by every lane
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (long long)b * N * I + hd * DH, 0, (int)((unsigned)(nq - 1) * (unsigned)I * 2u + 128u), 0x00020000);
    if (wave < nqt) {
      const int q = qt * 32 + li;
      float m = -INFINITY, l = 0.f;
      f32x16 o[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] = 0.f;
        o[1][r] = 0.f;
      }
      int kt = 0;
      for (; kt + 2 <= nkt; kt += 2) attn_key_tiles<2>(Ks, Vs, qf, m, l, o, sc, kt, nkt, N, li, h, lane, fsw);
      if (kt < nkt) attn_key_tiles<1>(Ks, Vs, qf, m, l, o, sc, kt, nkt, N, li, h, lane, fsw);
      const float inv = 1.f / l;
      const unsigned orow = (unsigned)q * (unsigned)I * 2u;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          fx4 v = {o[dt][4 * c] * inv, o[dt][4 * c + 1] * inv, o[dt][4 * c + 2] * inv, o[dt][4 * c + 3] * inv};
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, __builtin_convertvector(v, bf16x4)), ro, orow + (unsigned)(dt * 32 + 8 * c + 4 * h) * 2u, 0, 0);
        }
      if constexpr (HAS_LSE) {
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(lse + ((long long)b * H + hd) * N, 0, nq * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m + __builtin_amdgcn_logf(l)), rl, h == 0 ? (unsigned)q * 4u : 0x80000000u, 0, 0);
      }
    }
    __builtin_amdgcn_s_barrier();   // the next item is complete in LDS (the loader waited for it); this item's K / V buffers are free
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------ backward
// Two kernels, each one workgroup of 8 waves per (frame, head), probabilities recomputed from the saved base-2
// log-sum-exp (nothing of size N x N is stored):
//   dq kernel (wave = 32-query tile; K, V row images and K transposed in LDS):
//     P^T = exp2(K Q^T sc - lse), dP^T = V dO^T, delta = rowsum(dO o O), dS^T = P^T o (dP^T - delta) * scale,
//     dQ^T += K^T dS^T          (delta is also written out for the second kernel)
//   dkv kernel (wave = 32-key tile; Q, dO row images and both transposed in LDS):
//     P = exp2(Q K^T sc - lse), dP = dO V^T, dS = P o (dP - delta) * scale, dV^T += dO^T P, dK^T += Q^T dS
// Every product keeps the "accumulator registers are the next MFMA's B operand" orientation of the forward kernel.

// rows [0, N) of a 64-wide per-head column block -> LDS row image [NP][128 B], chunks swizzled by ((row >> 1) & 7).
// (staging loops request all of a thread's 16-byte loads before the first LDS write: a load-store-load-store loop would
//  serialise one memory round trip per chunk)
template <int NTHR>
__device__ __forceinline__ void stage_rows(unsigned char* img, const bf16_t* src, long long ld, int N, int NP, int tid) {
  constexpr int CH = 4;
  for (int f0 = tid; f0 < NP * 8; f0 += CH * NTHR) {
    u32x4_t v[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int f = f0 + j * NTHR, row = f >> 3, pc = f & 7;
      const bool ok = f < NP * 8 && row < N;
      v[j] = *reinterpret_cast<const u32x4_t*>(src + (ok ? row : 0) * ld + (pc ^ ((row >> 1) & 7)) * 8);
      if (!ok) v[j] = u32x4_t{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int f = f0 + j * NTHR;
      if (f < NP * 8) *reinterpret_cast<u32x4_t*>(img + (f >> 3) * 128 + (f & 7) * 16) = v[j];
    }
  }
}
// the same block transposed: img[d][vt_pos(row)], row stride VS elements (see the header of this file)
template <int NTHR>
__device__ __forceinline__ void stage_transposed(bf16_t* img, const bf16_t* src, long long ld, int N, int NP, int VS, int tid) {
  constexpr int CH = 2;
  const int total = (NP / 2) * 8;
  for (int f0 = tid; f0 < total; f0 += CH * NTHR) {
    u32x4_t v0[CH], v1[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int f = f0 + j * NTHR, row = (f >> 3) * 2, dc = f & 7;
      const bool ok0 = f < total && row < N, ok1 = f < total && row + 1 < N;
      v0[j] = *reinterpret_cast<const u32x4_t*>(src + (ok0 ? row : 0) * ld + dc * 8);
      v1[j] = *reinterpret_cast<const u32x4_t*>(src + (ok1 ? row + 1 : 0) * ld + dc * 8);
      if (!ok0) v0[j] = u32x4_t{0u, 0u, 0u, 0u};
      if (!ok1) v1[j] = u32x4_t{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int f = f0 + j * NTHR, row = (f >> 3) * 2, dc = f & 7;
      if (f < total) {
        unsigned* dst = reinterpret_cast<unsigned*>(img + (dc * 8) * VS + vt_pos(row));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dst[(2 * i) * (VS / 2)] = (v0[j][i] & 0xFFFFu) | (v1[j][i] << 16);
          dst[(2 * i + 1) * (VS / 2)] = (v0[j][i] >> 16) | (v1[j][i] & 0xFFFF0000u);
        }
      }
    }
  }
}

// acc += rows(img, tile base row `row0`) . frags   (A = 32 image rows x 64 deep, B = per-lane fragments)
__device__ __forceinline__ void mfma_rows(f32x16& acc, const unsigned char* img, int row0, int li, int h, const bf16x8 (&fb)[4]) {
  const unsigned fsw = (unsigned)((li >> 1) & 7);
  const unsigned char* rp = img + (row0 + li) * 128;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(rp + (((2 * s + h) ^ fsw) * 16));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fb[s], acc, 0, 0, 0);
  }
}
// acc[dt] += transposed(img)[d tile dt][32 contraction rows from `pos0`] . bf16(x)   (x = 32x32 fp32 tile, rows contracted)
__device__ __forceinline__ void mfma_transposed(f32x16 (&acc)[2], const bf16_t* img, int VS, int pos0, int li, int h, const f32x16& x) {
  bf16x8 xf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) xf[s][j] = (__bf16)x[8 * s + j];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(img + (dt * 32 + li) * VS + pos0 + 16 * s + 8 * h);
      acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[s], acc[dt], 0, 0, 0);
    }
}
__device__ __forceinline__ void load_frags(bf16x8 (&f)[4], const bf16_t* rowptr, int h) {
#pragma unroll
  for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const bf16x8*>(rowptr + 16 * s + 8 * h);
}
// transposed accumulator pair (rows = d, token on the lane) -> bf16 row `rowptr` (64 wide)
__device__ __forceinline__ void store_T_bf16(const f32x16 (&o)[2], bf16_t* rowptr, int h, float mul) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      fx4 v = {o[dt][4 * c] * mul, o[dt][4 * c + 1] * mul, o[dt][4 * c + 2] * mul, o[dt][4 * c + 3] * mul};
      *reinterpret_cast<bf16x4*>(rowptr + dt * 32 + 8 * c + 4 * h) = __builtin_convertvector(v, bf16x4);
    }
}

__global__ void __launch_bounds__(512) attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o_fwd,
                                                               const bf16_t* __restrict__ d_out, const float* __restrict__ lse,
                                                               bf16_t* __restrict__ dqkv, float* __restrict__ delta, int N, int H,
                                                               float scale) {
  constexpr int DH = 64, NTHR = 512, NW = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nkt = (N + 31) / 32, NP = nkt * 32, VS = NP + 8;
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + NP * 128;
  bf16_t* Kt = reinterpret_cast<bf16_t*>(smem + 2 * NP * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int I = H * DH;
  const long long ld = 3ll * I;
  const bf16_t* base = qkv + (long long)b * N * ld + hd * DH;
  const float sc = scale * DGVIT_LOG2E;
  stage_rows<NTHR>(Ks, base + I, ld, N, NP, tid);
  stage_rows<NTHR>(Vs, base + 2 * I, ld, N, NP, tid);
  stage_transposed<NTHR>(Kt, base + I, ld, N, NP, VS, tid);
  __syncthreads();
  for (int qt = wave; qt < nkt; qt += NW) {
    const int q = qt * 32 + li, qc = q < N ? q : 0;
    bf16x8 qf[4], dof[4], of[4];
    load_frags(qf, base + qc * ld, h);
    load_frags(dof, d_out + ((long long)b * N + qc) * I + hd * DH, h);
    load_frags(of, o_fwd + ((long long)b * N + qc) * I + hd * DH, h);
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += (float)dof[s][j] * (float)of[s][j];
    dl += __shfl_xor(dl, 32, 64);
    const float lq = lse[((long long)b * H + hd) * N + qc];
    f32x16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dq[0][r] = 0.f;
      dq[1][r] = 0.f;
    }
    for (int kt = 0; kt < nkt; ++kt) {
      f32x16 s0, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = 0.f;
        dp[r] = 0.f;
      }
      mfma_rows(s0, Ks, kt * 32, li, h, qf);
      mfma_rows(dp, Vs, kt * 32, li, h, dof);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + acc_row(r, h);
        const float pr = key < N ? __builtin_amdgcn_exp2f(s0[r] * sc - lq) : 0.f;
        s0[r] = pr * (dp[r] - dl) * scale;   // dS^T
      }
      mfma_transposed(dq, Kt, VS, kt * 32, li, h, s0);
    }
    if (q < N) {
      store_T_bf16(dq, dqkv + ((long long)b * N + q) * ld + hd * DH, h, 1.f);
      if (h == 0) delta[((long long)b * H + hd) * N + q] = dl;
    }
  }
}

__global__ void __launch_bounds__(512) attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_out,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                bf16_t* __restrict__ dqkv, int N, int H, float scale) {
  constexpr int DH = 64, NTHR = 512, NW = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nkt = (N + 31) / 32, NP = nkt * 32, VS = NP + 8;
  unsigned char* Qs = smem;
  unsigned char* Os = smem + NP * 128;                                     // dO rows
  bf16_t* Qt = reinterpret_cast<bf16_t*>(smem + 2 * NP * 128);
  bf16_t* Ot = Qt + 64 * VS;                                               // dO transposed
  float* lse_s = reinterpret_cast<float*>(Ot + 64 * VS);
  float* del_s = lse_s + NP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int I = H * DH;
  const long long ld = 3ll * I;
  const bf16_t* base = qkv + (long long)b * N * ld + hd * DH;
  const bf16_t* dob = d_out + (long long)b * N * I + hd * DH;
  const float sc = scale * DGVIT_LOG2E;
  stage_rows<NTHR>(Qs, base, ld, N, NP, tid);
  stage_rows<NTHR>(Os, dob, I, N, NP, tid);
  stage_transposed<NTHR>(Qt, base, ld, N, NP, VS, tid);
  stage_transposed<NTHR>(Ot, dob, I, N, NP, VS, tid);
  for (int i = tid; i < NP; i += NTHR) {
    lse_s[i] = i < N ? lse[((long long)b * H + hd) * N + i] : 0.f;
    del_s[i] = i < N ? delta[((long long)b * H + hd) * N + i] : 0.f;
  }
  __syncthreads();
  for (int kt = wave; kt < nkt; kt += NW) {
    const int key = kt * 32 + li, kc = key < N ? key : 0;
    const bool kvalid = key < N;
    bf16x8 kf[4], vf[4];
    load_frags(kf, base + I + kc * ld, h);
    load_frags(vf, base + 2 * I + kc * ld, h);
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f;
    }
    for (int qt = 0; qt < nkt; ++qt) {
      f32x16 s0, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = 0.f;
        dp[r] = 0.f;
      }
      mfma_rows(s0, Qs, qt * 32, li, h, kf);     // S[query][key]: queries in the registers, key on the lane
      mfma_rows(dp, Os, qt * 32, li, h, vf);     // dP = dO V^T
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int q0 = qt * 32 + 8 * g + 4 * h;  // registers 4g .. 4g+3 hold queries q0 .. q0+3
        const fx4 l4 = *reinterpret_cast<const fx4*>(lse_s + q0), d4 = *reinterpret_cast<const fx4*>(del_s + q0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * g + i;
          const float pr = (kvalid && q0 + i < N) ? __builtin_amdgcn_exp2f(s0[r] * sc - l4[i]) : 0.f;
          s0[r] = pr;                                // P
          dp[r] = pr * (dp[r] - d4[i]) * scale;      // dS
        }
      }
      mfma_transposed(dv, Ot, VS, qt * 32, li, h, s0);   // dV^T += dO^T P
      mfma_transposed(dk, Qt, VS, qt * 32, li, h, dp);   // dK^T += Q^T dS
    }
    if (kvalid) {
      store_T_bf16(dk, dqkv + ((long long)b * N + key) * ld + I + hd * DH, h, 1.f);
      store_T_bf16(dv, dqkv + ((long long)b * N + key) * ld + 2 * I + hd * DH, h, 1.f);
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
  const size_t lds = (size_t)2 * NP * 128;
  const float scale = 1.0f / sqrtf((float)dh);
  const double flops = 4.0 * (double)nq * N * dh * H * B;
  const int slot = profile_begin(PROF_ATTN_FWD, flops, st);
  // many items of more than four query tiles (BASELINE config 5: N = 197, 5280 items): the persistent kernel with LDS-DMA prefetch
  const long long items = (long long)B * H;
  if ((N + 31) / 32 > 4 && items >= 512 && 3ll * H * dh * 2 * 256 < (1ll << 31)) {
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    static int cached_cus = 0;
    if (!cached_cus) cached_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cus = cached_cus;
    static DeviceOnce once;
    if (const unsigned long long bit = once.pending()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_bf16_stream_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_bf16_stream_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return dgvit_set_error(DGVIT_ERR_HIP, "attention_fwd_bf16: cannot raise the dynamic LDS limit");
      once.mark(bit);
    }
    const unsigned grid = (unsigned)(items < cus ? items : cus);
    if (lse) hipLaunchKernelGGL((attn_fwd_bf16_stream_kernel<true>), dim3(grid), dim3(512), 160 * 1024, st, qkv, out, lse, N, H, scale, nq, (int)items);
    else hipLaunchKernelGGL((attn_fwd_bf16_stream_kernel<false>), dim3(grid), dim3(512), 160 * 1024, st, qkv, out, lse, N, H, scale, nq, (int)items);
  } else if ((nq + 31) / 32 > 4)
    hipLaunchKernelGGL((attn_fwd_bf16_kernel<8>), dim3((unsigned)((long long)B * H)), dim3(512), lds, st, qkv, out, lse, N, H, scale, nq);
  else
    hipLaunchKernelGGL((attn_fwd_bf16_kernel<4>), dim3((unsigned)((long long)B * H)), dim3(256), lds, st, qkv, out, lse, N, H, scale, nq);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("attn_fwd_bf16_kernel");
  return DGVIT_OK;
}

// dqkv (B, N, 3I) bf16 = gradient of the attention core; delta: B*H*N floats of scratch
int attention_bwd_bf16(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv, float* delta, int B,
                       int N, int H, int dh, hipStream_t st) {
  DGVIT_CHECK_ARG(qkv && out && dout && lse && dqkv && delta && B > 0 && H > 0, "attention_bwd_bf16: bad arguments");
  DGVIT_CHECK_ARG(dh == 64, "attention_bwd_bf16: dim_head=%d unsupported (64)", dh);
  DGVIT_CHECK_ARG(N >= 1 && N <= 224, "attention_bwd_bf16: tokens N=%d outside [1, 224]", N);
  const int NP = (N + 31) / 32 * 32;
  const size_t lds_q = (size_t)2 * NP * 128 + (size_t)64 * (NP + 8) * 2;
  const size_t lds_kv = (size_t)2 * NP * 128 + (size_t)2 * 64 * (NP + 8) * 2 + (size_t)2 * NP * 4;
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) !=
            hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "attention_bwd_bf16: cannot raise the dynamic LDS limit");
    once.mark(bit);
  }
  const float scale = 1.0f / sqrtf((float)dh);
  const double flops = 10.0 * (double)N * N * dh * H * B;   // 2.5 x forward
  const int slot = profile_begin(PROF_ATTN_BWD, flops, st);
  hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel, dim3((unsigned)((long long)B * H)), dim3(512), lds_q, st, qkv, out, dout, lse, dqkv, delta, N,
                     H, scale);
  hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel, dim3((unsigned)((long long)B * H)), dim3(512), lds_kv, st, qkv, dout, lse, delta, dqkv, N, H,
                     scale);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("attention_bwd_bf16");
  return DGVIT_OK;
}
