// HBM-bound helpers of the bf16 configuration: fp32 -> bf16 casts (weights), patch gather straight to bf16,
// LayerNorm reading the fp32 residual stream and writing the bf16 GEMM operand, bf16 transposes (backward operands).
#include "bf16.h"
#include "kernels.h"

namespace {

__global__ void __launch_bounds__(256) cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const fx4 v = reinterpret_cast<const fx4*>(src)[i];
  reinterpret_cast<bf16x4*>(dst)[i] = __builtin_convertvector(v, bf16x4);
}

// the same for patch widths that are multiples of 4: a thread moves four consecutive pixels of one patch row (16-byte load, 8-byte
// store; the one-pixel form ran at 1.5 TB/s on 224 x 224 frames).  idx4 walks the OUTPUT in units of four elements.
__global__ void __launch_bounds__(256) patchify_bf16_x4_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int Hi,
                                                               int Wi, int ph, int pw) {
  const int gw = Wi / pw, gh = Hi / ph, pd4 = ph * pw / 4, pw4 = pw / 4;
  const long long total4 = (long long)B * gh * gw * pd4;
  const long long idx4 = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx4 >= total4) return;
  const int e4 = (int)(idx4 % pd4);
  const long long bp = idx4 / pd4;
  const int p = (int)(bp % (gh * gw));
  const long long b = bp / (gh * gw);
  const int p1 = e4 / pw4, q = e4 % pw4, hy = p / gw, wx = p % gw;
  const fx4 v = *reinterpret_cast<const fx4*>(img + (b * Hi + hy * ph + p1) * Wi + wx * pw + 4 * q);
  reinterpret_cast<bf16x4*>(out)[idx4] = __builtin_convertvector(v, bf16x4);
}

// several casts in one launch (the weight arena: 4 matrices per layer + the patch embedding; one launch per matrix was 49 launches of
// ~5 us for the 0.1 ms of HBM time the 12-layer pack takes).  A block converts 1024 consecutive float4 of ONE segment: blocks are
// dealt to the segments through the prefix table.
__global__ void __launch_bounds__(256) cast_f32_bf16_batch_kernel(const CastBatch b) {
  int lo = 0, hi = b.nseg;      // last segment whose first block is <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((unsigned)b.first_block[mid] <= blockIdx.x) lo = mid; else hi = mid;
  }
  const long long n4 = b.n4[lo];
  const fx4* __restrict__ src = reinterpret_cast<const fx4*>(b.src[lo]);
  bf16x4* __restrict__ dst = reinterpret_cast<bf16x4*>(b.dst[lo]);
  const long long base = (long long)(blockIdx.x - (unsigned)b.first_block[lo]) * 1024 + threadIdx.x;
  fx4 v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = base + 256 * j < n4 ? src[base + 256 * j] : fx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + 256 * j < n4) dst[base + 256 * j] = __builtin_convertvector(v[j], bf16x4);
}

// 'b (h p1) (w p2) -> b (h w) (p1 p2)' (GoalFormer.py:138) with the cast to bf16 fused
__global__ void __launch_bounds__(256) patchify_bf16_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int Hi,
                                                            int Wi, int ph, int pw) {
  const int gw = Wi / pw, gh = Hi / ph, pd = ph * pw;
  const long long total = (long long)B * gh * gw * pd;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int e = (int)(idx % pd);
  const long long bp = idx / pd;
  const int p = (int)(bp % (gh * gw));
  const long long b = bp / (gh * gw);
  const int p1 = e / pw, p2 = e % pw, hy = p / gw, wx = p % gw;
  const __bf16 v = (__bf16)img[(b * Hi + hy * ph + p1) * Wi + wx * pw + p2];
  out[idx] = __builtin_bit_cast(bf16_t, v);
}

// nn.LayerNorm(D), eps 1e-5 (GoalFormer.py:34,37): fp32 row in, bf16 row out; one wave per row, row kept in registers.
// ADD: the row is first completed with the bf16 branch output of the previous sub-block (GoalFormer.py:103-104:
// x = attn(..) + x / x = ff(..) + x): v = x + delta, written back to the fp32 residual stream `xout` (unless NULL).  With `delta2`
// both branch outputs of a block join at once: v = (x + delta) + delta2 -- the no-grad forward does not store the stream between
// the attention and the feed-forward branch (22 instead of 24 bytes per element and block).
template <int NCH, int ADD>     // ADD: 0 plain, 1 x + delta, 2 (x + delta) + delta2
__global__ void __launch_bounds__(256) layernorm_fwd_bf16_kernel(const float* __restrict__ x, const bf16_t* __restrict__ delta,
                                                                 const bf16_t* __restrict__ delta2, float* __restrict__ xout,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                                 float* __restrict__ mean, float* __restrict__ rstd, int T, int D,
                                                                 float eps, int rs) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const float* xr = x + (long long)row * rs * D;
  fx4 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 4 + i * 256;
    v[i] = fx4{0.f, 0.f, 0.f, 0.f};
    if (c < D) {
      v[i] = *reinterpret_cast<const fx4*>(xr + c);
      if (ADD) {
        // (x + delta) + delta2 in this order: the sums a two-step schedule (x_mid written, then x_mid + delta2) would form.
        // (ADD is a template parameter: with a run-time test on delta2 inside this loop the loads of the later chunks were no longer
        //  issued ahead of the first use, and the three-stream kernel ran at 5.4 instead of 6.4 TB/s)
        const bf16x4 d1 = *reinterpret_cast<const bf16x4*>(delta + (long long)row * rs * D + c);
        bf16x4 d2 = d1;
        if (ADD == 2) d2 = *reinterpret_cast<const bf16x4*>(delta2 + (long long)row * rs * D + c);
        v[i] += __builtin_convertvector(d1, fx4);
        if (ADD == 2) v[i] += __builtin_convertvector(d2, fx4);
      }
    }
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  if (ADD && xout) {      // the completed row goes back to the residual stream (after every load of the row has been issued)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane * 4 + i * 256;
      if (c < D) *reinterpret_cast<fx4*>(xout + (long long)row * rs * D + c) = v[i];
    }
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < D) {
      const fx4 d = v[i] - mu;
      q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  const float rsd = rsqrtf(wave_sum(q) / (float)D + eps);
  bf16_t* yr = y + (long long)row * rs * D;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < D) {
      const fx4 g = *reinterpret_cast<const fx4*>(gamma + c);
      const fx4 b = *reinterpret_cast<const fx4*>(beta + c);
      const fx4 o = (v[i] - mu) * rsd * g + b;
      *reinterpret_cast<bf16x4*>(yr + c) = __builtin_convertvector(o, bf16x4);
    }
  }
  if (mean && lane == 0) {
    mean[row] = mu;
    rstd[row] = rsd;
  }
}

// xout = x + delta on `rows` rows of D (row r at r * rs * D): the last residual add of the encoder
__global__ void __launch_bounds__(256) residual_add_bf16_kernel(const float* __restrict__ x, const bf16_t* __restrict__ delta,
                                                                float* __restrict__ xout, long long rows, int D4, long long stride) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * D4) return;
  const long long r = i / D4, off = r * stride + (i % D4) * 4;
  *reinterpret_cast<fx4*>(xout + off) =
      *reinterpret_cast<const fx4*>(x + off) + __builtin_convertvector(*reinterpret_cast<const bf16x4*>(delta + off), fx4);
}

}  // namespace

int residual_add_bf16(const float* x, const bf16_t* delta, float* xout, int rows, int D, int rs, hipStream_t st) {
  DGVIT_CHECK_ARG(x && delta && xout && rows > 0 && D > 0 && D % 4 == 0, "residual_add_bf16: bad arguments");
  const long long n4 = (long long)rows * (D / 4);
  const int slot = profile_begin(PROF_OTHER, 0.0, st);
  hipLaunchKernelGGL(residual_add_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, delta, xout, (long long)rows,
                     D / 4, (long long)rs * D);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("residual_add_bf16");
  return DGVIT_OK;
}

int cast_f32_bf16(const float* src, bf16_t* dst, long long n, hipStream_t st) {
  DGVIT_CHECK_ARG(src && dst && n > 0 && n % 4 == 0, "cast_f32_bf16: n must be a positive multiple of 4");
  const long long n4 = n / 4;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, src, dst, n4);
  DGVIT_CHECK_LAUNCH("cast_f32_bf16");
  return DGVIT_OK;
}

void cast_batch_init(CastBatch& b) { b.nseg = 0; b.first_block[0] = 0; }

int cast_batch_add(CastBatch& b, const float* src, bf16_t* dst, long long n, hipStream_t st) {
  DGVIT_CHECK_ARG(src && dst && n > 0 && n % 4 == 0, "cast_f32_bf16: n must be a positive multiple of 4");
  if (b.nseg == DGVIT_CAST_SEGMENTS) {
    const int rc = cast_batch_flush(b, st);
    if (rc != DGVIT_OK) return rc;
  }
  const long long n4 = n / 4, blocks = (n4 + 1023) / 1024;
  DGVIT_CHECK_ARG(b.first_block[b.nseg] + blocks < (1ll << 31), "cast_f32_bf16: too many elements in one batch");
  b.src[b.nseg] = src; b.dst[b.nseg] = dst; b.n4[b.nseg] = n4;
  b.first_block[b.nseg + 1] = b.first_block[b.nseg] + (int)blocks;
  ++b.nseg;
  return DGVIT_OK;
}

int cast_batch_flush(CastBatch& b, hipStream_t st) {
  if (b.nseg == 0) return DGVIT_OK;
  hipLaunchKernelGGL(cast_f32_bf16_batch_kernel, dim3((unsigned)b.first_block[b.nseg]), dim3(256), 0, st, b);
  DGVIT_CHECK_LAUNCH("cast_f32_bf16_batch");
  cast_batch_init(b);
  return DGVIT_OK;
}

int patchify_bf16(const float* img, bf16_t* out, int B, int Hi, int Wi, int ph, int pw, hipStream_t st) {
  DGVIT_CHECK_ARG(img && out && B > 0, "patchify_bf16: bad arguments");
  DGVIT_CHECK_ARG(ph > 0 && pw > 0 && Hi % ph == 0 && Wi % pw == 0, "Image dimensions must be divisible by the patch size.");
  const long long total = (long long)B * Hi * Wi;
  if (pw % 4 == 0 && Wi % 4 == 0 && ((uintptr_t)img & 15) == 0 && ((uintptr_t)out & 7) == 0) {   // four pixels of a patch row per thread
    hipLaunchKernelGGL(patchify_bf16_x4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, img, out, B, Hi, Wi, ph, pw);
    DGVIT_CHECK_LAUNCH("patchify_bf16");
    return DGVIT_OK;
  }
  hipLaunchKernelGGL(patchify_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, img, out, B, Hi, Wi, ph, pw);
  DGVIT_CHECK_LAUNCH("patchify_bf16");
  return DGVIT_OK;
}

template <int ADD>
static int layernorm_launch(const float* x, const bf16_t* delta, const bf16_t* delta2, float* xout, const float* gamma, const float* beta,
                            bf16_t* y, float* mean, float* rstd, int T, int D, float eps, int rs, hipStream_t st) {
  DGVIT_CHECK_ARG(x && gamma && beta && y && T > 0, "layernorm_bf16: bad arguments");
  DGVIT_CHECK_ARG(D > 0 && D % 4 == 0 && D <= 1024, "layernorm_bf16: D=%d must be a multiple of 4 and <= 1024", D);
  const int slot = profile_begin(PROF_OTHER, 0.0, st);
  const dim3 grid((unsigned)((T + 3) / 4)), blk(256);
  if (D <= 256)
    hipLaunchKernelGGL((layernorm_fwd_bf16_kernel<1, ADD>), grid, blk, 0, st, x, delta, delta2, xout, gamma, beta, y, mean, rstd, T, D, eps, rs);
  else if (D <= 512)
    hipLaunchKernelGGL((layernorm_fwd_bf16_kernel<2, ADD>), grid, blk, 0, st, x, delta, delta2, xout, gamma, beta, y, mean, rstd, T, D, eps, rs);
  else if (D <= 768)     // (ViT-Base: no dead fourth chunk in the row loops)
    hipLaunchKernelGGL((layernorm_fwd_bf16_kernel<3, ADD>), grid, blk, 0, st, x, delta, delta2, xout, gamma, beta, y, mean, rstd, T, D, eps, rs);
  else
    hipLaunchKernelGGL((layernorm_fwd_bf16_kernel<4, ADD>), grid, blk, 0, st, x, delta, delta2, xout, gamma, beta, y, mean, rstd, T, D, eps, rs);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("layernorm_fwd_bf16");
  return DGVIT_OK;
}

int layernorm_fwd_bf16(const float* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd, int T, int D,
                       float eps, int rs, hipStream_t st) {
  return layernorm_launch<0>(x, nullptr, nullptr, nullptr, gamma, beta, y, mean, rstd, T, D, eps, rs, st);
}

int add_layernorm_fwd_bf16(const float* x, const bf16_t* delta, float* xout, const float* gamma, const float* beta, bf16_t* y,
                           float* mean, float* rstd, int T, int D, float eps, int rs, hipStream_t st) {
  DGVIT_CHECK_ARG(delta && xout, "add_layernorm_bf16: bad arguments");
  return layernorm_launch<1>(x, delta, nullptr, xout, gamma, beta, y, mean, rstd, T, D, eps, rs, st);
}

// y = LN((x + delta) + delta2), the sum written to xout unless it is NULL; delta2 may be NULL (no-grad forward, dgvit_api.hip)
int add2_layernorm_fwd_bf16(const float* x, const bf16_t* delta, const bf16_t* delta2, float* xout, const float* gamma, const float* beta,
                            bf16_t* y, int T, int D, float eps, hipStream_t st) {
  DGVIT_CHECK_ARG(delta, "add2_layernorm_bf16: bad arguments");
  if (delta2) return layernorm_launch<2>(x, delta, delta2, xout, gamma, beta, y, nullptr, nullptr, T, D, eps, 1, st);
  return layernorm_launch<1>(x, delta, nullptr, xout, gamma, beta, y, nullptr, nullptr, T, D, eps, 1, st);
}

// ---------------------------------------------------------------------------------------------- backward helpers
namespace {

// dst (cols x ldd) = src (rows x cols, row stride ld)^T on 64 x 64 tiles through LDS; dst columns [rows, ldd) are zero
// filled (ldd = rows rounded up to 8: the token dimension of the weight-gradient operands).  SRC = bf16_t or float
// (the fp32 -> bf16 cast of a weight fused with its transposition).  cols % 8 == 0.
typedef bf16_t u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4m __attribute__((ext_vector_type(4)));

// dst (cols x ldd) = src (rows x cols, row stride ld)^T on 64 x 64 tiles (ldd >= rows, multiple of 8; columns beyond `rows`
// are zero filled).  SRC = float: the fp32 -> bf16 cast of a weight fused with its transposition (the B operands of the
// data-gradient GEMMs); a bf16 source works the same way.  cols % 8 == 0.
// A thread loads 8 columns of TWO adjacent rows and interleaves them in registers into 8 dwords {row r, row r + 1} of one
// column each -- already transposed pairs -- so the LDS tile T[col][row pair] is written with dword stores (stride 33:
// at most 2-way bank conflicts) and read back for the output rows with one ds_read_b128 per 16-byte store.
template <typename SRC>
__device__ __forceinline__ u32x4m load8(const SRC* p) {
  if constexpr (sizeof(SRC) == 2) {
    return *reinterpret_cast<const u32x4m*>(p);
  } else {
    const fx4 a = *reinterpret_cast<const fx4*>(p), b = *reinterpret_cast<const fx4*>(p + 4);
    const bf16x4 a4 = __builtin_convertvector(a, bf16x4), b4 = __builtin_convertvector(b, bf16x4);
    // (whole-vector shuffle + bit cast: element-wise writes into a 16-bit vector were miscompiled by hipcc 7.2)
    return __builtin_bit_cast(u32x4m, __builtin_shufflevector(a4, b4, 0, 1, 2, 3, 4, 5, 6, 7));
  }
}

template <typename SRC>
__global__ void __launch_bounds__(256) transpose_bf16_kernel(const SRC* __restrict__ src, long long ld, bf16_t* __restrict__ dst,
                                                             int rows, int cols, int ldd) {
  __shared__ unsigned tile[64][33];   // [column][row pair]
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  {
    const int rp = tid >> 3, c8 = (tid & 7) * 8, r = r0 + 2 * rp;   // 32 row pairs x 8 column chunks
    u32x4m lo = {0u, 0u, 0u, 0u}, hi = {0u, 0u, 0u, 0u};
    if (c0 + c8 < cols) {
      if (r < rows) lo = load8(src + (long long)r * ld + c0 + c8);
      if (r + 1 < rows) hi = load8(src + (long long)(r + 1) * ld + c0 + c8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tile[c8 + 2 * i][rp] = (lo[i] & 0xFFFFu) | (hi[i] << 16);
      tile[c8 + 2 * i + 1][rp] = (lo[i] >> 16) | (hi[i] & 0xFFFF0000u);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + i * 256, c = f >> 3, r8 = (f & 7) * 8;   // output row c0 + c, output columns r0 + r8 .. + 7
    if (c0 + c < cols && r0 + r8 < ldd) {
      u32x4m o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = tile[c][(r8 >> 1) + j];   // rows >= `rows` were staged as zeros
      *reinterpret_cast<u32x4m*>(dst + (long long)(c0 + c) * ldd + r0 + r8) = o;
    }
  }
}

}  // namespace

// ---- column sums of a bf16 matrix (bias gradient db[n] = sum_t dY[t][n]): row-block partials, then colpart_reduce
namespace {
// grid (ceil(cols / 512), nblk); 256 threads = 64 column chunks of 8 x 4 row lanes; partial[blockIdx.y][col]
__global__ void __launch_bounds__(256) colsum_bf16_kernel(const bf16_t* __restrict__ src, long long ld, float* __restrict__ part, int rows,
                                                          int cols, int rows_per_blk) {
  __shared__ float red[4][512];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, c = blockIdx.x * 512 + tx * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    const int r1 = min(rows, (int)(blockIdx.y + 1) * rows_per_blk);
    int r = blockIdx.y * rows_per_blk + ty;
    for (; r + 12 < r1; r += 16) {   // 4 rows in flight per thread
      bf16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(src + (long long)(r + 4 * u) * ld + c);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (float)v[u][j];
    }
    for (; r < r1; r += 4) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (long long)r * ld + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ty][tx * 8 + j] = s[j];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256)
    if (blockIdx.x * 512 + i < cols)
      part[(long long)blockIdx.y * cols + blockIdx.x * 512 + i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
}
}  // namespace

int colsum_bf16_blocks(int rows) { return rows < 4096 ? 1 : (rows < 16384 ? 64 : 256); }

namespace {
// out[c] = sum over the nblk row-block partials of column c; 64 columns x 16 row lanes per workgroup, fixed order
__global__ void __launch_bounds__(1024) colpart_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int cols) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, c = blockIdx.x * 64 + tx;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int z = ty;
    for (; z + 16 < nblk; z += 32) {
      s0 += part[(long long)z * cols + c];
      s1 += part[(long long)(z + 16) * cols + c];
    }
    if (z < nblk) s0 += part[(long long)z * cols + c];
  }
  red[ty][tx] = s0 + s1;
  __syncthreads();
  if (ty == 0 && c < cols) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][tx];
    out[c] = s;
  }
}
}  // namespace

// out (cols) = sum of nblk partial rows
static int colpart_reduce_n(const float* part, float* out, int nblk, int cols, hipStream_t st) {
  DGVIT_CHECK_ARG(part && out && nblk > 0 && cols > 0, "colpart_reduce: bad arguments");
  const int slot = profile_begin(PROF_OTHER, 0.0, st);
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((cols + 63) / 64), dim3(1024), 0, st, part, out, nblk, cols);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("colpart_reduce");
  return DGVIT_OK;
}

// out[c] = sum_r src[r][c]; part: colsum_bf16_blocks(rows) * cols floats
int colsum_bf16(const bf16_t* src, long long ld, float* out, float* part, int rows, int cols, hipStream_t st) {
  DGVIT_CHECK_ARG(src && out && part && rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0, "colsum_bf16: cols and ld must be multiples of 8");
  const int nblk = colsum_bf16_blocks(rows), rpb = (rows + nblk - 1) / nblk;
  const int slot = profile_begin(PROF_OTHER, 0.0, st);
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((cols + 511) / 512, nblk), dim3(256), 0, st, src, ld, part, rows, cols, rpb);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("colsum_bf16");
  return colpart_reduce_n(part, out, nblk, cols, st);
}

int transpose_cast_f32_bf16(const float* src, bf16_t* dst, int rows, int cols, hipStream_t st) {
  DGVIT_CHECK_ARG(src && dst && rows > 0 && cols > 0 && cols % 8 == 0 && rows % 8 == 0,
                  "transpose_cast_f32_bf16: rows and cols must be multiples of 8");
  hipLaunchKernelGGL((transpose_bf16_kernel<float>), dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, st, src, (long long)cols, dst,
                     rows, cols, rows);
  DGVIT_CHECK_LAUNCH("transpose_cast_f32_bf16");
  return DGVIT_OK;
}
