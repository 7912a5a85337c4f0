// Single-workgroup MFMA helpers (v_mfma_f32_32x32x2_f32) shared by the latency-bound kernels: the fused SAC heads (heads.hip)
// and the per-frame encoder kernels of the small-batch path (frame.hip).  One operand always sits in an LDS image; the other
// is either a weight matrix read straight from L2 or a second LDS image.
#pragma once
#include "common.h"

constexpr int HR = 32;          // rows of one MFMA tile

__device__ __forceinline__ int arow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// A single workgroup walks the whole head, so every global round trip that is not overlapped is pure latency: the weight
// fragments of LB consecutive 8-deep k-groups are requested together (LB or 4 * LB loads in flight per lane) before the MFMAs
// that consume them.
constexpr int LB = 8;

// acc(32x32) += A B with A rows in an LDS image (k contiguous, stride sa) and B[k][j = lane] = W[(jb + li) * ldw + k]
// (weight rows straight from global / L2; k beyond kmax and rows beyond nmax read as 0)
template <bool VEC>
__device__ __forceinline__ void mm_rows_x_wrows(f32x16& acc, const float* a_img, int sa, const float* __restrict__ W, int ldw, int jn,
                                                int nmax, int kmax, int KP, int li, int h) {
  const bool jok = jn < nmax;
  const float* wrow = W + (long long)(jok ? jn : 0) * ldw;
  const int ng = KP / 8;
  for (int g0 = 0; g0 < ng; g0 += LB) {
    float4 b[LB];
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      const int k = 8 * (g0 + u) + 4 * h;
      if (VEC) {
        b[u] = (jok && g0 + u < ng && k < kmax) ? *reinterpret_cast<const float4*>(wrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (((ldw | kmax) & 1) == 0 && (reinterpret_cast<uintptr_t>(W) & 7) == 0) {
        // rows 8-byte aligned, even K (the CNN heads' 290 inputs): two 8-byte loads instead of four scalar ones
        const float2 lo = (jok && g0 + u < ng && k + 0 < kmax) ? *reinterpret_cast<const float2*>(wrow + k) : make_float2(0.f, 0.f);
        const float2 hi = (jok && g0 + u < ng && k + 2 < kmax) ? *reinterpret_cast<const float2*>(wrow + k + 2) : make_float2(0.f, 0.f);
        b[u] = make_float4(lo.x, lo.y, hi.x, hi.y);
      } else {
        b[u].x = (jok && g0 + u < ng && k + 0 < kmax) ? wrow[k + 0] : 0.f;
        b[u].y = (jok && g0 + u < ng && k + 1 < kmax) ? wrow[k + 1] : 0.f;
        b[u].z = (jok && g0 + u < ng && k + 2 < kmax) ? wrow[k + 2] : 0.f;
        b[u].w = (jok && g0 + u < ng && k + 3 < kmax) ? wrow[k + 3] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      if (g0 + u < ng) {
        const float4 a = *reinterpret_cast<const float4*>(a_img + li * sa + 8 * (g0 + u) + 4 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[u].w, acc, 0, 0, 0);
      }
    }
  }
}

// acc(32x32) += A B with A rows in LDS (k contiguous) and B[k][j = lane] = W[k * ldw + jb + li]  (k = the weight's ROW index:
// lanes run along its contiguous dimension); k < kmax, j < jmax
__device__ __forceinline__ void mm_rows_x_wcols(f32x16& acc, const float* a_img, int sa, const float* __restrict__ W, int ldw, int jn,
                                                int jmax, int kmax, int li, int h) {
  const bool jok = jn < jmax;
  const int ng = (kmax + 7) / 8;
  for (int g0 = 0; g0 < ng; g0 += LB) {
    float b[LB][4];
#pragma unroll
    for (int u = 0; u < LB; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 8 * (g0 + u) + 4 * h + s;
        b[u][s] = (jok && k < kmax) ? W[(long long)k * ldw + jn] : 0.f;
      }
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      if (g0 + u < ng) {
        const float4 a = *reinterpret_cast<const float4*>(a_img + li * sa + 8 * (g0 + u) + 4 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[u][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[u][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[u][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[u][3], acc, 0, 0, 0);
      }
    }
  }
}

// acc(32x32) += A^T B over the HR = 32 batch rows: A[i = lane][k = row] = imgA[row * sa + ib + li], B[k = row][j = lane] = imgB[row * sb + jb + li]
__device__ __forceinline__ void mm_cols_x_cols(f32x16& acc, const float* imgA, int sa, int ib, const float* imgB, int sb, int jb, int li, int h) {
#pragma unroll
  for (int g = 0; g < 32 / 8; ++g)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int row = 8 * g + 4 * h + s;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(imgA[row * sa + ib + li], imgB[row * sb + jb + li], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

