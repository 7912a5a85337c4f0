// Fused MLP heads of the SAC networks (got_sac_network.py:114-121, 230-235, 433-435 and the CNN twins :157-166, 303-307):
//     y_j = W3_j relu(W2 relu(W1 cat(x_0, x_1, x_2) + b1) + b2) + b3_j          j < heads3, for 1 or 2 independent towers
// as ONE launch forward and ONE launch backward (+ one grouped reduction when the batch spans several workgroups), instead
// of a GEMM / ReLU-mask / split-K / reduction launch per Linear.  A policy head is one tower with two third layers
// (mean_linear, log_std_linear); a twin-Q head is two towers (fc1,fc2,fc3 | fc11,fc21,fc31) over the same concatenated
// input.  These heads are < 0.1 % of the model's FLOPs but were ~40 of the ~230 launches of a training step, and at the
// reference's shipped batch of 32 every launch is pure latency.
//
// One workgroup = 32 batch rows of one tower, 4 waves, fp32 MFMA (v_mfma_f32_32x32x2_f32), activations in LDS:
//   forward : x rows -> LDS; layer 1 and 2 as 32x32 output blocks per wave (A = LDS rows, B = weight rows straight from L2),
//             bias + ReLU in registers, h1 / h2 to LDS (next layer's A) and to HBM (backward); layer 3 (<= 4 outputs) on the VALU.
//   backward: dh2 = relu'(h2) o sum_j dy_j W3_j (VALU); dW3, db3; dW2 = dh2^T h1 and dW1 = dh1^T x contract over the 32 rows
//             (both operands read down LDS columns); dh1 = relu'(h1) o dh2 W2 and dx = dh1 W1 contract over features
//             (B = weight rows, lanes along the contiguous output index).  Parameter-gradient partials of workgroup w go to
//             scratch[w] and are summed in a fixed order (deterministic); a single-workgroup batch writes the gradients directly.
#include "common.h"
#include "kernels.h"
#include "small_mma.h"

namespace {

constexpr int HMAXN = 128;      // widest hidden layer
constexpr int HMAXK = 512;      // widest concatenated input

struct HeadArgs {
  int B, nseg, kx[3], ldx[3], K0, KP;      // KP = K0 rounded up to 8 (LDS image width without the pad)
  int n1, n2, n3, towers, heads3;
  const float* x[3];
  const float* w1[2]; const float* b1[2]; const float* w2[2]; const float* b2[2];
  const float* w3[2][2]; const float* b3[2][2];
  float* h1; float* h2; float* y;           // (towers, B, n1) (towers, B, n2) (towers, heads3, B, n3)
  // backward
  const float* dyp[2][2];                   // per (tower, third layer): (B, n3) gradient of its output, null = no gradient
  float* dx[3]; int lddx[3];                // per input piece, may be null; towers ACCUMULATE into it (tower 1 adds to tower 0's)
  float* part;                              // [nwg][towers] partial parameter gradients, layout see part_offsets
  long long part_stride;                    // floats per (workgroup, tower)
  int direct;                               // 1: a single row block -> write parameter gradients straight to dparams
  float* dx1;                               // twin head: tower 1 leaves its input gradient here ([B][K0]); head_dx_add_kernel adds it to dx
  float* dw1[2]; float* db1[2]; float* dw2[2]; float* db2[2]; float* dw3[2][2]; float* db3[2][2];
};

// concatenated input rows [row0, row0 + 32) -> LDS image [32][KP + 4], zero beyond B and beyond K0
__device__ __forceinline__ void stage_x(const HeadArgs& a, float* xs, int row0, int tid) {
  const int SX = a.KP + 4;
  int koff = 0;
  for (int sg = 0; sg < a.nseg; ++sg) {
    const int kx = a.kx[sg], ld = a.ldx[sg];
    const float* __restrict__ src = a.x[sg];
    // 16 loads in flight per thread and trip (the whole 32 x 290 input of the CNN heads in three trips): the kernel is one
    // workgroup walking a chain of dependent global round trips, every trip saved is ~1.5 us
    for (int f0 = tid; f0 < HR * kx; f0 += 256 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int f = f0 + u * 256, r = f / kx, k = f - r * kx;
        v[u] = (f < HR * kx && row0 + r < a.B) ? src[(long long)(row0 + r) * ld + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int f = f0 + u * 256, r = f / kx, k = f - r * kx;
        if (f < HR * kx) xs[r * SX + koff + k] = v[u];
      }
    }
    koff += kx;
  }
  const int pad = a.KP - a.K0;
  for (int f = tid; f < HR * pad; f += 256) xs[(f / pad) * SX + a.K0 + f % pad] = 0.f;
}

// ------------------------------------------------------------------------------------------------ forward
template <bool VEC>
__global__ void __launch_bounds__(256) mlp_head_fwd_kernel(const HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int SX = a.KP + 4, S1 = a.n1 + 4, S2 = a.n2 + 4;
  float* xs = smem;
  float* h1s = xs + HR * SX;
  float* h2s = h1s + HR * S1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.x * HR, t = blockIdx.y;
  stage_x(a, xs, row0, tid);
  __syncthreads();
  // layer 1
  for (int cb = wave; cb * 32 < a.n1; cb += 4) {
    f32x16 acc;
    zero16(acc);
    const int n = cb * 32 + li;
    mm_rows_x_wrows<VEC>(acc, xs, SX, a.w1[t], a.K0, n, a.n1, a.K0, a.KP, li, h);
    const float bias = n < a.n1 ? a.b1[t][n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = arow(r, h);
      const float v = fmaxf(acc[r] + bias, 0.f);
      if (n < a.n1) {
        h1s[row * S1 + n] = v;
        if (row0 + row < a.B) a.h1[((long long)t * a.B + row0 + row) * a.n1 + n] = v;
      }
    }
  }
  __syncthreads();
  // layer 2
  for (int cb = wave; cb * 32 < a.n2; cb += 4) {
    f32x16 acc;
    zero16(acc);
    const int n = cb * 32 + li;
    mm_rows_x_wrows<true>(acc, h1s, S1, a.w2[t], a.n1, n, a.n2, a.n1, a.n1, li, h);
    const float bias = n < a.n2 ? a.b2[t][n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = arow(r, h);
      const float v = fmaxf(acc[r] + bias, 0.f);
      if (n < a.n2) {
        h2s[row * S2 + n] = v;
        if (row0 + row < a.B) a.h2[((long long)t * a.B + row0 + row) * a.n2 + n] = v;
      }
    }
  }
  __syncthreads();
  // layer 3: one thread per (third layer j, row, output o)
  for (int f = tid; f < a.heads3 * HR * a.n3; f += 256) {
    const int o = f % a.n3, row = (f / a.n3) % HR, j = f / (a.n3 * HR);
    if (row0 + row >= a.B) continue;
    const float* w = a.w3[t][j] + (long long)o * a.n2;
    float s = a.b3[t][j][o];
#pragma unroll 8
    for (int k = 0; k < a.n2; k += 4) {   // n2 % 32 == 0; W3 rows are 16-byte aligned when W3 is
      const float4 wv = *reinterpret_cast<const float4*>(w + k), hv = *reinterpret_cast<const float4*>(h2s + row * S2 + k);
      s = fmaf(hv.x, wv.x, fmaf(hv.y, wv.y, fmaf(hv.z, wv.z, fmaf(hv.w, wv.w, s))));
    }
    a.y[(((long long)t * a.heads3 + j) * a.B + row0 + row) * a.n3 + o] = s;
  }
}

// ------------------------------------------------------------------------------------------------ backward
// partial layout per (workgroup, tower): [dw1 n1*K0 | db1 n1 | dw2 n2*n1 | db2 n2 | heads3 x (dw3 n3*n2 | db3 n3)], each padded to 4
struct PartOff { long long w1, b1, w2, b2, w3[2], b3[2], total; };
__host__ __device__ inline long long pad4(long long n) { return (n + 3) & ~3ll; }
__host__ __device__ inline PartOff part_offsets(int K0, int n1, int n2, int n3, int heads3) {
  PartOff o;
  long long p = 0;
  o.w1 = p; p += pad4((long long)n1 * K0);
  o.b1 = p; p += pad4(n1);
  o.w2 = p; p += pad4((long long)n2 * n1);
  o.b2 = p; p += pad4(n2);
  for (int j = 0; j < 2; ++j) {
    o.w3[j] = p; if (j < heads3) p += pad4((long long)n3 * n2);
    o.b3[j] = p; if (j < heads3) p += pad4(n3);
  }
  o.total = p;
  return o;
}

template <bool VEC>
__global__ void __launch_bounds__(256) mlp_head_bwd_kernel(const HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int SX = a.KP + 4, S1 = a.n1 + 4, S2 = a.n2 + 4;
  float* xs = smem;                    // [32][KP+4]    input rows
  float* h1s = xs + HR * SX;           // [32][n1+4]    relu(layer 1), later dh1
  float* h2s = h1s + HR * S1;          // [32][n2+4]    relu(layer 2)
  float* g2s = h2s + HR * S2;          // [32][n2+4]    dh2
  float* g1s = g2s + HR * S2;          // [32][n1+4]    dh1
  float* dys = g1s + HR * S1;          // [heads3][32][n3]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.x * HR, t = blockIdx.y;
  const PartOff po = part_offsets(a.K0, a.n1, a.n2, a.n3, a.heads3);
  float* part = a.part + ((long long)blockIdx.x * a.towers + t) * a.part_stride;
  float* o_w1 = a.direct ? a.dw1[t] : part + po.w1;
  float* o_b1 = a.direct ? a.db1[t] : part + po.b1;
  float* o_w2 = a.direct ? a.dw2[t] : part + po.w2;
  float* o_b2 = a.direct ? a.db2[t] : part + po.b2;

  stage_x(a, xs, row0, tid);
  // h1 / h2 rows (n1, n2 multiples of 32: whole float4 pieces) -> LDS: every piece of a thread is requested before the first is used,
  // from a clamped row (rows past the batch are zeroed afterwards).  The per-element form this replaces compiled to one load and one
  // `s_waitcnt vmcnt(0)` per element -- 32 dependent L2 round trips in front of the first MFMA of a kernel that is one workgroup's
  // latency chain (round 4: 69 -> 4x us at batch 512).
  auto stage_rows = [&](float* dst, int SD, const float* __restrict__ src, int n) {
    const int n4 = n >> 2, total = HR * n4;
    for (int f0 = tid; f0 < total; f0 += 256 * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int f = f0 + u * 256 < total ? f0 + u * 256 : 0, r = f / n4, c = (f - r * n4) * 4;
        const int row = row0 + r < a.B ? row0 + r : a.B - 1;
        v[u] = *reinterpret_cast<const float4*>(src + ((long long)t * a.B + row) * n + c);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int f = f0 + u * 256;
        if (f < total) {
          const int r = f / n4, c = (f - r * n4) * 4;
          *reinterpret_cast<float4*>(dst + r * SD + c) = row0 + r < a.B ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  };
  stage_rows(h1s, S1, a.h1, a.n1);
  stage_rows(h2s, S2, a.h2, a.n2);
  // third-layer weights W3_j (n3 x n2 each, <= 2 x 4 x 128 floats) -> LDS: the dh2 loop below reads every element 32 times
  float* w3s = dys + a.heads3 * HR * a.n3;
  for (int f = tid; f < a.heads3 * a.n3 * a.n2; f += 256) {
    const int j = f / (a.n3 * a.n2);
    w3s[f] = (j ? a.w3[t][1] : a.w3[t][0])[f - j * a.n3 * a.n2];
  }
  for (int f = tid; f < a.heads3 * HR * a.n3; f += 256) {
    const int o = f % a.n3, row = (f / a.n3) % HR, j = f / (a.n3 * HR);
    const float* dyj = j ? a.dyp[t][1] : a.dyp[t][0];   // (a per-lane index into a kernel-argument array would go through scratch memory)
    const int rc = row0 + row < a.B ? row0 + row : a.B - 1;
    const float v = dyj ? dyj[(long long)rc * a.n3 + o] : 0.f;
    dys[f] = row0 + row < a.B ? v : 0.f;
  }
  __syncthreads();
  // dh2 = relu'(h2) o sum_j dy_j W3_j
  for (int f = tid; f < HR * a.n2; f += 256) {
    const int r = f / a.n2, c = f % a.n2;
    float s = 0.f;
    for (int j = 0; j < a.heads3; ++j)
      for (int o = 0; o < a.n3; ++o) s = fmaf(dys[(j * HR + r) * a.n3 + o], w3s[(j * a.n3 + o) * a.n2 + c], s);
    g2s[r * S2 + c] = h2s[r * S2 + c] > 0.f ? s : 0.f;
  }
  // dW3_j[o][c] = sum_rows dy_j[row][o] h2[row][c];  db3_j[o] = sum_rows dy_j[row][o]
  for (int f = tid; f < a.heads3 * a.n3 * (a.n2 + 1); f += 256) {
    const int c = f % (a.n2 + 1), o = (f / (a.n2 + 1)) % a.n3, j = f / ((a.n2 + 1) * a.n3);
    float s = 0.f;
    for (int r = 0; r < HR; ++r) s = fmaf(dys[(j * HR + r) * a.n3 + o], c < a.n2 ? h2s[r * S2 + c] : 1.f, s);
    const bool used = (j ? a.dyp[t][1] : a.dyp[t][0]) != nullptr;     // an unused third layer gets no gradient
    float* ow = !used ? nullptr : a.direct ? (j ? a.dw3[t][1] : a.dw3[t][0]) : part + (j ? po.w3[1] : po.w3[0]);
    float* ob = !used ? nullptr : a.direct ? (j ? a.db3[t][1] : a.db3[t][0]) : part + (j ? po.b3[1] : po.b3[0]);
    if (c < a.n2) { if (ow) ow[(long long)o * a.n2 + c] = s; }
    else if (ob) ob[o] = s;
  }
  __syncthreads();
  // db2 = column sums of dh2
  if (o_b2)
    for (int c = tid; c < a.n2; c += 256) {
      float s = 0.f;
      for (int r = 0; r < HR; ++r) s += g2s[r * S2 + c];
      o_b2[c] = s;
    }
  // dW2 (n2 x n1) = dh2^T h1, 32x32 blocks round-robin over the waves
  if (o_w2) {
    const int nb1 = a.n1 / 32, nb2 = a.n2 / 32;
    for (int blk = wave; blk < nb1 * nb2; blk += 4) {
      const int ib = (blk / nb1) * 32, jb = (blk % nb1) * 32;
      f32x16 acc;
      zero16(acc);
      mm_cols_x_cols(acc, g2s, S2, ib, h1s, S1, jb, li, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) o_w2[(long long)(ib + arow(r, h)) * a.n1 + jb + li] = acc[r];
    }
  }
  // dh1 = relu'(h1) o (dh2 W2): contraction over n2, lanes along W2's contiguous n1
  for (int cb = wave; cb * 32 < a.n1; cb += 4) {
    f32x16 acc;
    zero16(acc);
    const int n = cb * 32 + li;
    mm_rows_x_wcols(acc, g2s, S2, a.w2[t], a.n1, n, a.n1, a.n2, li, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = arow(r, h);
      g1s[row * S1 + n] = h1s[row * S1 + n] > 0.f ? acc[r] : 0.f;
    }
  }
  __syncthreads();
  if (o_b1)
    for (int c = tid; c < a.n1; c += 256) {
      float s = 0.f;
      for (int r = 0; r < HR; ++r) s += g1s[r * S1 + c];
      o_b1[c] = s;
    }
  // dW1 (n1 x K0) = dh1^T x
  if (o_w1) {
    const int nb1 = a.n1 / 32, nbk = (a.K0 + 31) / 32;
    for (int blk = wave; blk < nb1 * nbk; blk += 4) {
      const int ib = (blk / nbk) * 32, jb = (blk % nbk) * 32;
      f32x16 acc;
      zero16(acc);
      // x image columns beyond KP are not staged: clamp the lane's column (its result is not stored)
      const int jl = jb + li < a.KP ? li : 0;
      mm_cols_x_cols(acc, g1s, S1, ib, xs, SX, jb + jl - li, li, h);
      if (jb + li < a.K0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o_w1[(long long)(ib + arow(r, h)) * a.K0 + jb + li] = acc[r];
      }
    }
  }
  // dx = dh1 W1: contraction over n1, lanes along W1's contiguous K0; scattered into the input pieces
  bool any_dx = false;
  for (int sg = 0; sg < a.nseg; ++sg) any_dx = any_dx || a.dx[sg];
  if (any_dx) {
    for (int cb = wave; cb * 32 < a.K0; cb += 4) {
      const int k = cb * 32 + li;
      f32x16 acc;
      zero16(acc);
      mm_rows_x_wcols(acc, g1s, S1, a.w1[t], a.K0, k, a.K0, a.n1, li, h);
      if (k < a.K0) {
        // which input piece column k belongs to: selected with ?: (a per-lane index into the kernel-argument arrays is a load from
        // memory, and inside the store loop below it was one dependent round trip per accumulator register: 16 in a row)
        const int k0 = a.kx[0], k1 = a.nseg > 1 ? a.kx[1] : 0;
        const int sg = (a.nseg > 1 && k >= k0) ? ((a.nseg > 2 && k >= k0 + k1) ? 2 : 1) : 0;
        const int kk = sg == 0 ? k : (sg == 1 ? k - k0 : k - k0 - k1);
        float* d = sg == 0 ? a.dx[0] : (sg == 1 ? a.dx[1] : a.dx[2]);
        const long long ldd = sg == 0 ? a.lddx[0] : (sg == 1 ? a.lddx[1] : a.lddx[2]);
        // both towers of a twin head run in this launch: tower 0 writes dx, tower 1 its own buffer, added afterwards in a fixed order
        const bool own = t == 1 && a.dx1;
        float* dst = !d ? nullptr : (own ? a.dx1 + k : d + kk);       // (a piece without a gradient slot gets none from either tower)
        const long long ldo = own ? a.K0 : ldd;
        if (dst) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row0 + arow(r, h);
            if (row < a.B) dst[(long long)row * ldo] = acc[r];
          }
        }
      }
    }
  }
}

// dx += dx1 (tower 1's input gradient of a twin head), scattered into the input pieces
__global__ void __launch_bounds__(256) head_dx_add_kernel(const HeadArgs a) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)a.B * a.K0) return;
  const int row = (int)(idx / a.K0);
  int kk = (int)(idx - (long long)row * a.K0), sg = 0;
  while (sg + 1 < a.nseg && kk >= a.kx[sg]) { kk -= a.kx[sg]; ++sg; }
  float* d = a.dx[sg];
  if (d) d[(long long)row * a.lddx[sg] + kk] += a.dx1[idx];
}

size_t fwd_lds(int KP, int n1, int n2) { return sizeof(float) * HR * ((size_t)(KP + 4) + (n1 + 4) + (n2 + 4)); }
size_t bwd_lds(int KP, int n1, int n2, int n3, int heads3) {
  return sizeof(float) * (HR * ((size_t)(KP + 4) + 2 * (n1 + 4) + 2 * (n2 + 4)) + (size_t)heads3 * HR * n3 + (size_t)heads3 * n3 * n2 + 4);
}

int fill_args(HeadArgs& a, const dgvit_mlp_desc* d, const float* const* in, const float* const* params) {
  DGVIT_CHECK_ARG(d && in && params, "mlp_head: null pointer");
  DGVIT_CHECK_ARG(d->batch > 0 && d->nseg >= 1 && d->nseg <= 3, "mlp_head: batch must be positive, 1..3 input pieces");
  DGVIT_CHECK_ARG(d->towers >= 1 && d->towers <= 2 && d->heads3 >= 1 && d->heads3 <= 2, "mlp_head: 1 or 2 towers / third layers");
  DGVIT_CHECK_ARG(d->n1 > 0 && d->n1 <= HMAXN && d->n1 % 32 == 0 && d->n2 > 0 && d->n2 <= HMAXN && d->n2 % 32 == 0 && d->n3 >= 1 && d->n3 <= 4,
                  "mlp_head: hidden widths must be multiples of 32 up to %d, at most 4 outputs", HMAXN);
  a = HeadArgs{};
  a.B = d->batch; a.nseg = d->nseg;
  a.K0 = 0;
  for (int s = 0; s < d->nseg; ++s) {
    DGVIT_CHECK_ARG(in[s] && d->kx[s] > 0 && d->ldx[s] >= d->kx[s], "mlp_head: bad input piece %d", s);
    a.x[s] = in[s]; a.kx[s] = d->kx[s]; a.ldx[s] = d->ldx[s];
    a.K0 += d->kx[s];
  }
  DGVIT_CHECK_ARG(a.K0 <= HMAXK, "mlp_head: concatenated input wider than %d", HMAXK);
  a.KP = (a.K0 + 7) & ~7;
  a.n1 = d->n1; a.n2 = d->n2; a.n3 = d->n3; a.towers = d->towers; a.heads3 = d->heads3;
  const int per = 4 + 2 * d->heads3;
  for (int t = 0; t < d->towers; ++t) {
    const float* const* p = params + t * per;
    for (int i = 0; i < per; ++i) DGVIT_CHECK_ARG(p[i], "mlp_head: parameter %d of tower %d is null", i, t);
    a.w1[t] = p[0]; a.b1[t] = p[1]; a.w2[t] = p[2]; a.b2[t] = p[3];
    DGVIT_CHECK_ARG((reinterpret_cast<uintptr_t>(p[2]) & 15) == 0, "mlp_head: layer-2 weight of tower %d must be 16-byte aligned", t);
    for (int j = 0; j < d->heads3; ++j) {
      a.w3[t][j] = p[4 + 2 * j]; a.b3[t][j] = p[5 + 2 * j];
      DGVIT_CHECK_ARG((reinterpret_cast<uintptr_t>(p[4 + 2 * j]) & 15) == 0, "mlp_head: third-layer weight %d of tower %d must be 16-byte aligned", j, t);
    }
  }
  return DGVIT_OK;
}

bool w1_vec(const HeadArgs& a) {
  bool v = a.K0 % 4 == 0;
  for (int t = 0; t < a.towers; ++t) v = v && (reinterpret_cast<uintptr_t>(a.w1[t]) & 15) == 0;
  return v;
}

}  // namespace

int mlp_head_forward(const dgvit_mlp_desc* d, const float* const* in, const float* const* params, float* h1, float* h2, float* y,
                     hipStream_t stream) {
  HeadArgs a;
  int rc = fill_args(a, d, in, params);
  if (rc) return rc;
  DGVIT_CHECK_ARG(h1 && h2 && y, "mlp_head_forward: null output");
  a.h1 = h1; a.h2 = h2; a.y = y;
  const size_t lds = fwd_lds(a.KP, a.n1, a.n2);
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    const int mx = (int)fwd_lds(HMAXK, HMAXN, HMAXN);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_head_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_head_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "mlp_head_forward: hipFuncSetAttribute failed");
    once.mark(bit);
  }
  const dim3 grid((a.B + HR - 1) / HR, a.towers);
  const int slot = profile_begin(PROF_OTHER, 0.0, stream);
  if (w1_vec(a)) hipLaunchKernelGGL(mlp_head_fwd_kernel<true>, grid, dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(mlp_head_fwd_kernel<false>, grid, dim3(256), lds, stream, a);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("mlp_head_forward");
  return DGVIT_OK;
}

long long mlp_head_backward_scratch(const dgvit_mlp_desc* d) {
  if (!d || d->batch <= 0 || d->nseg < 1 || d->nseg > 3) return -1;
  int K0 = 0;
  for (int s = 0; s < d->nseg; ++s) K0 += d->kx[s];
  const int nwg = (d->batch + HR - 1) / HR;
  const long long dx1 = d->towers == 2 ? (((long long)d->batch * K0 + 3) & ~3ll) : 0;   // tower 1's input gradient of a twin head
  if (nwg == 1) return 4 + dx1;
  return (long long)nwg * d->towers * part_offsets(K0, d->n1, d->n2, d->n3, d->heads3).total + dx1;
}

int mlp_head_backward(const dgvit_mlp_desc* d, const float* const* in, const float* const* params, const float* h1, const float* h2,
                      const float* const* dy, float* const* din, float* const* dparams, float* scratch, long long scratch_floats,
                      hipStream_t stream) {
  HeadArgs a;
  int rc = fill_args(a, d, in, params);
  if (rc) return rc;
  DGVIT_CHECK_ARG(h1 && h2 && dy && din && dparams, "mlp_head_backward: null pointer");
  const long long need = mlp_head_backward_scratch(d);
  DGVIT_CHECK_ARG(scratch && scratch_floats >= need, "mlp_head_backward: scratch %lld < %lld floats", scratch_floats, need);
  a.h1 = const_cast<float*>(h1); a.h2 = const_cast<float*>(h2);
  for (int t = 0; t < a.towers; ++t)
    for (int j = 0; j < a.heads3; ++j) a.dyp[t][j] = dy[t * a.heads3 + j];
  for (int s = 0; s < a.nseg; ++s) { a.dx[s] = din[s]; a.lddx[s] = a.kx[s]; }
  const int per = 4 + 2 * a.heads3;
  for (int t = 0; t < a.towers; ++t) {
    float* const* g = dparams + t * per;
    a.dw1[t] = g[0]; a.db1[t] = g[1]; a.dw2[t] = g[2]; a.db2[t] = g[3];
    for (int j = 0; j < a.heads3; ++j) { a.dw3[t][j] = g[4 + 2 * j]; a.db3[t][j] = g[5 + 2 * j]; }
  }
  const int nwg = (a.B + HR - 1) / HR;
  const PartOff po = part_offsets(a.K0, a.n1, a.n2, a.n3, a.heads3);
  a.direct = nwg == 1;
  a.part = scratch; a.part_stride = po.total;
  const size_t lds = bwd_lds(a.KP, a.n1, a.n2, a.n3, a.heads3);
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    const int mx = (int)bwd_lds(HMAXK, HMAXN, HMAXN, 4, 2);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_head_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_head_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "mlp_head_backward: hipFuncSetAttribute failed");
    once.mark(bit);
  }
  // a twin head runs both towers in ONE launch (blockIdx.y): tower 0 writes dx, tower 1 writes its input gradient to scratch and a
  // small kernel adds it (x = t0 + t1: the order is fixed); the two towers used to be two launches one behind the other
  bool any_dx = false;
  for (int s = 0; s < a.nseg; ++s) any_dx = any_dx || a.dx[s];
  const bool vec = w1_vec(a);
  const int slot = profile_begin(PROF_OTHER, 0.0, stream);
  a.dx1 = (a.towers == 2 && any_dx) ? scratch + (need - (((long long)a.B * a.K0 + 3) & ~3ll)) : nullptr;
  if (vec) hipLaunchKernelGGL(mlp_head_bwd_kernel<true>, dim3(nwg, a.towers), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(mlp_head_bwd_kernel<false>, dim3(nwg, a.towers), dim3(256), lds, stream, a);
  if (a.dx1) {
    const long long n = (long long)a.B * a.K0;
    hipLaunchKernelGGL(head_dx_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  }
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("mlp_head_backward");
  if (a.direct) return DGVIT_OK;
  // fixed-order sums of the per-workgroup partials: one grouped launch per 8 tensors
  ReduceGroup g;
  reduce_group_init(g);
  const long long stride = po.total * a.towers;
  for (int t = 0; t < a.towers; ++t) {
    const float* base = scratch + (long long)t * po.total;
    auto add = [&](long long off, float* out, long long n) -> int {
      if (!out) return DGVIT_OK;
      return reduce_group_add(g, base + off, out, n, nullptr, n, nwg, stride, stream);
    };
    if ((rc = add(po.w1, a.dw1[t], (long long)a.n1 * a.K0))) return rc;
    if ((rc = add(po.b1, a.db1[t], a.n1))) return rc;
    if ((rc = add(po.w2, a.dw2[t], (long long)a.n2 * a.n1))) return rc;
    if ((rc = add(po.b2, a.db2[t], a.n2))) return rc;
    for (int j = 0; j < a.heads3; ++j) {
      if (!a.dyp[t][j]) continue;
      if ((rc = add(po.w3[j], a.dw3[t][j], (long long)a.n3 * a.n2))) return rc;
      if ((rc = add(po.b3[j], a.db3[t][j], a.n3))) return rc;
    }
  }
  return reduce_group_flush(g, stream);
}

// ------------------------------------------------------------------------------------------------ tanh-Gaussian sampling
// GoTPolicy.sample / GaussianPolicy.sample (got_sac_network.py:238-251, 310-321) as one launch each way instead of ~20 / ~30
// (B, 2)-sized elementwise launches:
//   ls = clamp(log_std_raw, ls_min, ls_max);  std = exp(ls);  x = mean + std * eps;  y = tanh(x)
//   action = y * scale + bias;   tanh_mean = tanh(mean) * scale + bias
//   log_prob = sum_a [ -eps^2 / 2 - ls - log(sqrt(2 pi)) - log(scale * (1 - y^2) + 1e-6) ]
// (Normal(mean, std).log_prob(x) evaluated at x = mean + std * eps is -eps^2/2 - ls - log sqrt(2 pi) exactly.)
namespace {
constexpr float kHalfLog2Pi = 0.91893853320467274178f;
constexpr float kEpsilon = 1e-6f;

__global__ void __launch_bounds__(256) tanh_gaussian_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ lsr,
                                                                const float* __restrict__ eps, const float* __restrict__ scale,
                                                                const float* __restrict__ bias, int sn, float lo, float hi,
                                                                float* __restrict__ action, float* __restrict__ logp,
                                                                float* __restrict__ tmean, int B, int A) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float lp = 0.f;
  for (int a = 0; a < A; ++a) {
    const int i = b * A + a;
    const float sc = scale[sn == 1 ? 0 : a], bi = bias[sn == 1 ? 0 : a];
    const float ls = fminf(fmaxf(lsr[i], lo), hi), e = eps[i], m = mean[i];
    const float y = tanhf(fmaf(expf(ls), e, m));
    action[i] = fmaf(y, sc, bi);
    tmean[i] = fmaf(tanhf(m), sc, bi);
    lp += -0.5f * e * e - ls - kHalfLog2Pi - logf(fmaf(sc, 1.f - y * y, kEpsilon));
  }
  logp[b] = lp;
}

// dmean, dlog_std_raw from the gradients of action / log_prob / tanh_mean (each may be null = zero)
__global__ void __launch_bounds__(256) tanh_gaussian_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ lsr,
                                                                const float* __restrict__ eps, const float* __restrict__ scale, int sn,
                                                                float lo, float hi, const float* __restrict__ dact,
                                                                const float* __restrict__ dlp, const float* __restrict__ dtm,
                                                                float* __restrict__ dmean, float* __restrict__ dls, int B, int A) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float gl = dlp ? dlp[b] : 0.f;
  for (int a = 0; a < A; ++a) {
    const int i = b * A + a;
    const float sc = scale[sn == 1 ? 0 : a];
    const float raw = lsr[i], ls = fminf(fmaxf(raw, lo), hi), e = eps[i], m = mean[i];
    const float sd = expf(ls);
    const float y = tanhf(fmaf(sd, e, m)), omy = 1.f - y * y;
    // d/dx of  action * dact  +  log_prob * gl   (x = mean + std * eps)
    const float dx = (dact ? dact[i] : 0.f) * sc * omy + gl * 2.f * y * sc * omy / fmaf(sc, omy, kEpsilon);
    const float tm = tanhf(m);
    dmean[i] = dx + (dtm ? dtm[i] : 0.f) * sc * (1.f - tm * tm);
    const float g = dx * sd * e - gl;                       // x depends on ls through std; log_prob has the explicit -ls
    dls[i] = (raw >= lo && raw <= hi) ? g : 0.f;            // torch.clamp passes the gradient inside [min, max] (bounds included)
  }
}
}  // namespace

int tanh_gaussian_forward(const float* mean, const float* lsr, const float* eps, const float* scale, const float* bias, int sn, float lo,
                          float hi, float* action, float* logp, float* tmean, int B, int A, hipStream_t stream) {
  DGVIT_CHECK_ARG(mean && lsr && eps && scale && bias && action && logp && tmean, "tanh_gaussian_forward: null pointer");
  DGVIT_CHECK_ARG(B > 0 && A > 0 && A <= 16 && (sn == 1 || sn == A) && lo <= hi, "tanh_gaussian_forward: bad sizes");
  hipLaunchKernelGGL(tanh_gaussian_fwd_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, mean, lsr, eps, scale, bias, sn, lo, hi, action,
                     logp, tmean, B, A);
  DGVIT_CHECK_LAUNCH("tanh_gaussian_forward");
  return DGVIT_OK;
}

int tanh_gaussian_backward(const float* mean, const float* lsr, const float* eps, const float* scale, int sn, float lo, float hi,
                           const float* dact, const float* dlp, const float* dtm, float* dmean, float* dls, int B, int A,
                           hipStream_t stream) {
  DGVIT_CHECK_ARG(mean && lsr && eps && scale && dmean && dls, "tanh_gaussian_backward: null pointer");
  DGVIT_CHECK_ARG(B > 0 && A > 0 && A <= 16 && (sn == 1 || sn == A) && lo <= hi, "tanh_gaussian_backward: bad sizes");
  hipLaunchKernelGGL(tanh_gaussian_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, mean, lsr, eps, scale, sn, lo, hi, dact, dlp,
                     dtm, dmean, dls, B, A);
  DGVIT_CHECK_LAUNCH("tanh_gaussian_backward");
  return DGVIT_OK;
}
