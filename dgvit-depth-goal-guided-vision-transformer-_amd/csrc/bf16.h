// bf16-storage / fp32-accumulate kernels of the DGViT encoder (BASELINE config 5: 224x224 ViT-Base variant).
// Internal declarations; the public C ABI is include/dgvit_hip.h.
#pragma once
#include "common.h"

typedef unsigned short bf16_t;   // raw bf16 bits in memory
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float fx4 __attribute__((ext_vector_type(4)));

// GELU for outputs that are rounded to bf16: x * sigmoid(x (c0 + c1 x^2 + c2 x^4)) with the coefficients fitted (minimax over |x| <= 12) to the
// erf form x Phi(x): |difference| <= 2.6e-5 everywhere (a bf16 ulp at 1.0 is 7.8e-3), exact 0 at 0 and the exact limits x and -0 in the tails
// (x^2 is clamped at 64, beyond which the sigmoid has saturated).  5 packed + 1 plain + 4 transcendental VALU instructions per PAIR of values
// against ~28 + 4 for two gelu_erf (common.h): the bf16 fc1 epilogue is VALU-bound, and this halves it.  The coefficients carry the
// -log2(e) of exp(-z) = exp2(-z log2 e).  fp32 outputs keep gelu_erf (5e-7).  tests/test_gpu_bf16.py::test_gelu_epilogue_against_the_erf_form
__device__ __forceinline__ fx4 gelu_bf16x4(fx4 x) {
  fx4 x2 = x * x;
  x2 = __builtin_elementwise_min(x2, fx4{64.f, 64.f, 64.f, 64.f});
  fx4 p = x2 * 0.0010142630198970437f + -0.10677572339773178f;
  p = p * x2 + -2.301121234893799f;
  const fx4 z = x * p;
  fx4 d;
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_exp2f(z[i]);
  d = d + 1.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_rcpf(d[i]);
  return x * d;
}

// C = A B^T with A (M,K) and B (N,K) bf16, k contiguous (the forward `Y = X W^T` shape; the backward GEMMs are
// brought into this shape by transposed bf16 copies of their operands), fp32 accumulate on v_mfma_f32_32x32x16_bf16.
enum GemmBf16Epilogue {
  BEPI_BF16 = 0,       // C (bf16) = acc + bias
  BEPI_GELU_BF16 = 1,  // C (bf16) = gelu_erf(acc + bias)
  BEPI_F32 = 2,        // C (fp32) = acc + bias + res       (residual stream stays fp32)
  BEPI_DGELU_BF16 = 3, // C (bf16) = acc * gelu'(aux)       aux (bf16) = saved pre-activation
  BEPI_F32_PLAIN = 4,  // C (fp32) = acc                    (weight gradients)
  BEPI_GELU2_BF16 = 5  // C (bf16) = gelu_erf(acc + bias);  C2 (bf16) = acc + bias (pre-activation kept for training)
};

struct GemmBf16Params {
  const bf16_t* A; int lda;
  const bf16_t* B; int ldb;
  int M, N, K;
  void* C; int ldc;
  const float* bias;            // [N] fp32 or null
  const float* res; int ldr;    // fp32 residual or null
  int res_mod;                  // > 0: residual row = (m % res_mod) + 1 (positional embedding)
  int c_rgrp;                   // > 0: physical C row = m + m / c_rgrp + 1 (patch rows -> token rows)
  bf16_t* C2; int ldc2;         // BEPI_GELU2_BF16: pre-activation copy
  const bf16_t* aux; int ldaux; // BEPI_DGELU_BF16
  int ksplit, kchunk;           // ksplit > 1: K is cut into ksplit slices of kchunk (multiple of 32); slice z writes C + z * slab_stride
  long long slab_stride;        //             (BEPI_F32_PLAIN only; the slabs are summed by reduce_slabs)
  int group_m;                  // row panels per walk group (0 = default); see tile_mn in gemm_bf16_ring_kernel
  int col_blocks;
  int slice_major;              // ring kernel, split-K: virtual tiles k-slice major (1) or tile major (0, the order until round 4)               // stream kernel: column blocks of the tile walk (set at launch; 0 = the older group_m walk)
  int stagger; long long stagger_cycles;   // stream kernel: start phases of the workgroups and cycles between them (set at launch)
#ifdef DGVIT_DIAG
  long long* diag_stamps;   // stream kernel, timing variant 512: [workgroup][wave][tile < 8][8] s_memtime stamps (tools/bf16_stream_stamps.py)
#endif
  int tn;                       // 1: A is (K, M) with row stride lda, B is (K, N): C = A^T B (BEPI_F32_PLAIN, 256 x 256 tile)
};

int gemm_bf16(int epi, const GemmBf16Params& p, hipStream_t st);
// gemm_bf16_stream.hip: the 64-deep / 128-byte-line kernel for the plain NT forms (bias, GELU, fp32 out); gemm_bf16 dispatches to it
bool gemm_bf16_stream_supports(int epi, const GemmBf16Params& p);
int gemm_bf16_stream(int epi, const GemmBf16Params& p, hipStream_t st);

int cast_f32_bf16(const float* src, bf16_t* dst, long long n, hipStream_t st);
// several fp32 -> bf16 casts as ONE launch (weight packing): add segments, flush launches them (and a full table flushes itself)
#define DGVIT_CAST_SEGMENTS 64
struct CastBatch {
  const float* src[DGVIT_CAST_SEGMENTS];
  bf16_t* dst[DGVIT_CAST_SEGMENTS];
  long long n4[DGVIT_CAST_SEGMENTS];
  int first_block[DGVIT_CAST_SEGMENTS + 1];
  int nseg;
};
void cast_batch_init(CastBatch& b);
int cast_batch_add(CastBatch& b, const float* src, bf16_t* dst, long long n, hipStream_t st);
int cast_batch_flush(CastBatch& b, hipStream_t st);
int patchify_bf16(const float* img, bf16_t* patches, int B, int ih, int iw, int ph, int pw, hipStream_t st);
int layernorm_fwd_bf16(const float* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd, int T, int D,
                       float eps, int rs, hipStream_t st);
int add_layernorm_fwd_bf16(const float* x, const bf16_t* delta, float* xout, const float* gamma, const float* beta, bf16_t* y,
                           float* mean, float* rstd, int T, int D, float eps, int rs, hipStream_t st);
int add2_layernorm_fwd_bf16(const float* x, const bf16_t* delta, const bf16_t* delta2, float* xout, const float* gamma, const float* beta,
                            bf16_t* y, int T, int D, float eps, hipStream_t st);
int residual_add_bf16(const float* x, const bf16_t* delta, float* xout, int rows, int D, int rs, hipStream_t st);
int layernorm_bwd_bf16(const bf16_t* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                       float* dx, bf16_t* dxb, float* dgamma, float* dbeta, float* partial, int T, int D, int rs, hipStream_t st);
int colsum_bf16_blocks(int rows);
int colsum_bf16(const bf16_t* src, long long ld, float* out, float* part, int rows, int cols, hipStream_t st);
int transpose_cast_f32_bf16(const float* src, bf16_t* dst, int rows, int cols, hipStream_t st);
int attention_fwd_bf16(const bf16_t* qkv, bf16_t* out, float* lse, int B, int N, int H, int dh, int nq, hipStream_t st);

int attention_bwd_bf16(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv, float* delta, int B,
                       int N, int H, int dh, hipStream_t st);

