// Shader clock under load: every wave issues back-to-back fp32 MFMAs (4 independent accumulators) and reads both the shader-clock
// counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around the loop.
//   shader MHz            = d(s_memtime) / d(s_memrealtime) * 100
//   pipe use              = MFMA cycles issued / d(s_memtime)
//   TFLOP/s               = flops / wall time of the launch (HIP events)
// Build: hipcc --offload-arch=gfx950 -O3 tools/native/clock_probe.hip -o tools/native/clock_probe
// Run:   tools/native/clock_probe [waves_per_simd=1] [iters=20000] [workgroups=256*waves_per_simd... computed]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>   // 0: v_mfma_f32_32x32x2_f32   1: v_mfma_f32_32x32x16_bf16   2: no MFMA (v_fma chain: a light kernel)
                      // 3: fp32 MFMA with RANDOM, changing operands (the data path toggles like a real GEMM's)
                      // 4: as 3, operands re-read from LDS at the fp32 GEMM's rate (3 ds_read_b128 per 8 MFMAs)
                      // 5: as 4 plus the GEMM's LDS write and global (L2-resident) read traffic: 1.5 b128 loads + ds_writes per 8 MFMAs
__global__ void __launch_bounds__(256) probe(long long* out, float* sink, int iters, const float4* __restrict__ gsrc) {
  __shared__ float lds[4096 + 2048];   // 24 KB: up to 6 workgroups per CU
  for (int i = threadIdx.x; i < 4096 + 2048; i += 256) lds[i] = (float)((i * 2654435761u) >> 9) * (1.0f / 4194304.0f) - 1.0f;
  __syncthreads();
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) a0[r] = a1[r] = a2[r] = a3[r] = 0.f;
  const float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  bf16x8 bx, by;
  for (int i = 0; i < 8; ++i) { bx[i] = (__bf16)x; by[i] = (__bf16)y; }
  float rx[8], ry[8];
  {
    unsigned st = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 12345u);
    for (int i = 0; i < 8; ++i) {
      st = st * 1664525u + 1013904223u; rx[i] = (float)(int)(st >> 8) * (1.0f / 8388608.0f) - 1.0f;
      st = st * 1664525u + 1013904223u; ry[i] = (float)(int)(st >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
  }
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(gsrc), 0, 0x400000, 0x00020000);
  float4 pre0 = make_float4(0.f, 0.f, 0.f, 0.f), pre1 = pre0, pre2 = pre0;
  float f = x;
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
  const long long tc0 = __builtin_readcyclecounter();      // s_memtime: shader clock
  const long long w0 = wall_clock64();                    // s_memrealtime: 100 MHz
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    } else if (KIND == 1) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a3, 0, 0, 0);
    } else if (KIND == 4 || KIND == 5) {
      const float4 qa = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + (i & 7) * 1040) & 4092));
      const float4 qb = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + 2080 + (i & 7) * 1040) & 4092));
      const float4 qc = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + 1040 + (i & 7) * 1040) & 4092));
      if (KIND == 5) {
        const float4 g0 = gsrc[(threadIdx.x + (i & 1023) * 256 + blockIdx.x * 64) & 0x3FFFF];
        float4 g1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i & 1) g1 = gsrc[(threadIdx.x + (i & 1023) * 256 + blockIdx.x * 64 + 131072) & 0x3FFFF];
        *reinterpret_cast<float4*>(lds + 4096 + ((threadIdx.x * 4 + (i & 1) * 1024) & 2044)) = make_float4(g0.x + g1.x, g0.y + g1.y, g0.z, g0.w);
      }
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, qb.x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, qc.x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, qb.y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, qc.y, a3, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, qb.z, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, qc.z, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, qb.w, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, qc.w, a3, 0, 0, 0);
    } else if (KIND == 11 || KIND == 12) {
      // as 9, plus the k-tile's operand fetch from an L2-resident buffer: 11 = LDS-DMA (buffer_load ... lds: no VGPRs, no ds_write),
      // 12 = the GEMM's way (16-byte loads into registers, ds_write_b128 one k-tile later); 3 x 16 bytes per thread per 16 MFMAs
      const float4 qa = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + (i & 7) * 1040) & 4092));
      const float4 qb = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + 2080 + (i & 7) * 1040) & 4092));
      const float4 qc = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + 1040 + (i & 7) * 1040) & 4092));
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, qb.x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, qc.x, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, qb.y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, qc.y, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, qb.z, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, qc.z, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, qb.w, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, qc.w, a1, 0, 0, 0);
      if (i & 1) {
        const unsigned go = (unsigned)(((threadIdx.x + (i & 1023) * 256 + blockIdx.x * 64) & 0x3FFFF) * 16);
        if (KIND == 11) {
          typedef __attribute__((address_space(3))) void* lds_ptr_t;
          float* dst = lds + 4096 + (threadIdx.x >> 6) * 256 + ((i >> 1) & 1) * 1024;   // wave-uniform base; lane l lands at +16 l bytes
          __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (lds_ptr_t)dst, 16, go, 0, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (lds_ptr_t)dst, 16, go ^ 0x100000u, 0, 0, 0);
          if (threadIdx.x < 128) __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (lds_ptr_t)dst, 16, go ^ 0x200000u, 0, 0, 0);
          asm volatile("s_waitcnt vmcnt(3)" ::: "memory");      // the previous k-tile's DMA has landed
        } else {
          *reinterpret_cast<float4*>(lds + 4096 + ((threadIdx.x * 4) & 2044)) = pre0;               // last k-tile's registers -> LDS
          *reinterpret_cast<float4*>(lds + 4096 + ((threadIdx.x * 4 + 1024) & 2044)) = pre1;
          if (threadIdx.x < 128) *reinterpret_cast<float4*>(lds + 4096 + ((threadIdx.x * 4 + 512) & 2044)) = pre2;
          pre0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs, go, 0, 0));
          pre1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs, go ^ 0x100000u, 0, 0));
          if (threadIdx.x < 128) pre2 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs, go ^ 0x200000u, 0, 0));
        }
        __syncthreads();
      }
    } else if (KIND == 9 || KIND == 10) {
      // as 8, plus what a k-tile of the GEMM adds: one workgroup barrier per 16 MFMAs (9), and the 3 LDS tile writes (10)
      const float4 qa = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + (i & 7) * 1040) & 4092));
      const float4 qb = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + 2080 + (i & 7) * 1040) & 4092));
      const float4 qc = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + 1040 + (i & 7) * 1040) & 4092));
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, qb.x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, qc.x, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, qb.y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, qc.y, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, qb.z, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, qc.z, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, qb.w, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, qc.w, a1, 0, 0, 0);
      if (i & 1) {
        if (KIND == 10) {
          *reinterpret_cast<float4*>(lds + 4096 + ((threadIdx.x * 4) & 2044)) = make_float4(qa.x, qb.y, qc.z, qa.w);
          *reinterpret_cast<float4*>(lds + 4096 + ((threadIdx.x * 4 + 1024) & 2044)) = make_float4(qb.x, qc.y, qa.z, qb.w);
          if (threadIdx.x < 128) *reinterpret_cast<float4*>(lds + 4096 + ((threadIdx.x * 4 + 512) & 2044)) = make_float4(qc.x, qa.y, qb.z, qc.w);
        }
        __syncthreads();
      }
    } else if (KIND == 8) {
      // as 4 with TWO accumulators (the fp32 GEMM's 32 x 64 wave tile): fits 5 workgroups per CU like the real kernel
      const float4 qa = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + (i & 7) * 1040) & 4092));
      const float4 qb = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + 2080 + (i & 7) * 1040) & 4092));
      const float4 qc = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + 1040 + (i & 7) * 1040) & 4092));
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, qb.x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, qc.x, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, qb.y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, qc.y, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, qb.z, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, qc.z, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, qb.w, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, qc.w, a1, 0, 0, 0);
    } else if (KIND == 6 || KIND == 7) {
      // v_mfma_f32_16x16x4_f32 (8 passes, 4 accumulator registers): the same FLOPs per cycle with half the accumulator traffic
      float4 qa, qb, qc;
      if (KIND == 7) {
        qa = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + (i & 7) * 1040) & 4092));
        qb = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + 2080 + (i & 7) * 1040) & 4092));
        qc = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + 1040 + (i & 7) * 1040) & 4092));
      } else {
        qa = make_float4(rx[0], rx[1], rx[2], rx[3]); qb = make_float4(ry[0], ry[1], ry[2], ry[3]); qc = make_float4(ry[4], ry[5], ry[6], ry[7]);
      }
      const float ea[4] = {qa.x, qa.y, qa.z, qa.w}, eb[4] = {qb.x, qb.y, qb.z, qb.w}, ec[4] = {qc.x, qc.y, qc.z, qc.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[e], eb[e], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[e], ec[e], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(eb[e], ec[e], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(ec[e], ea[e], c3, 0, 0, 0);
      }
    } else if (KIND == 3) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(rx[0], ry[0], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(rx[1], ry[1], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(rx[2], ry[2], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(rx[3], ry[3], a3, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(rx[4], ry[4], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(rx[5], ry[5], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(rx[6], ry[6], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(rx[7], ry[7], a3, 0, 0, 0);
    } else {
      f = __builtin_fmaf(f, y, x);
      f = __builtin_fmaf(f, y, x);
      f = __builtin_fmaf(f, y, x);
      f = __builtin_fmaf(f, y, x);
    }
  }
  const long long tc1 = __builtin_readcyclecounter();
  const long long w1 = wall_clock64();
  float s = f + pre0.x + pre1.y + pre2.z;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) {
    const int wv = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    out[2 * wv] = tc1 - tc0;
    out[2 * wv + 1] = w1 - w0;
  }
}

static float4* g_src = nullptr;

template <int KIND>
void run(const char* name, int wgs, int iters, double flops_per_mfma, int cycles_per_mfma) {
  const int waves = wgs * 4;
  long long* d;
  float* sink;
  hipMalloc(&d, sizeof(long long) * 2 * waves);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(wgs), dim3(256), 0, 0, d, sink, iters, g_src);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * waves);
    hipMemcpy(h.data(), d, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost);
    std::vector<double> mhz(waves), use(waves);
    for (int i = 0; i < waves; ++i) {
      mhz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
      use[i] = ((KIND >= 8) ? 8.0 : (KIND >= 6 ? 16.0 : (KIND >= 3 ? 8.0 : 4.0))) * iters * cycles_per_mfma / (double)h[2 * i];
    }
    std::sort(mhz.begin(), mhz.end());
    std::sort(use.begin(), use.end());
    const double per_iter = (KIND >= 8) ? 8.0 : (KIND >= 6 ? 16.0 : (KIND >= 3 ? 8.0 : 4.0));
    const double tf = KIND == 2 ? 0.0 : per_iter * iters * flops_per_mfma * waves / (ms * 1e-3) / 1e12;
    printf("%-28s wgs %5d iters %6d  %8.3f ms  shader clock MHz min/med/max %6.0f %6.0f %6.0f   pipe use med %.3f   %7.1f TFLOP/s\n", name, wgs, iters,
           ms, mhz.front(), mhz[waves / 2], mhz.back(), use[waves / 2], tf);
  }
  hipFree(d);
  hipFree(sink);
}

int main(int argc, char** argv) {
  const int per_simd = argc > 1 ? atoi(argv[1]) : 1;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device %s, %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int wgs = p.multiProcessorCount * per_simd;
  (void)hipMalloc(&g_src, sizeof(float4) * 0x40000);      // 4 MB: stays in L2 / MALL
  (void)hipMemset(g_src, 0, sizeof(float4) * 0x40000);
  run<2>("v_fma chain (light)", wgs, iters * 4, 0, 4);
  run<0>("v_mfma_f32_32x32x2_f32", wgs, iters, 2.0 * 32 * 32 * 2, 64);
  run<3>("f32 MFMA, random operands", wgs, iters / 2, 2.0 * 32 * 32 * 2, 64);
  run<4>("f32 MFMA + LDS operand reads", wgs, iters / 2, 2.0 * 32 * 32 * 2, 64);
  run<5>("f32 MFMA + LDS + L2 reads", wgs, iters / 2, 2.0 * 32 * 32 * 2, 64);
  run<8>("32x32x2, 2 accumulators + LDS reads", wgs, iters / 2, 2.0 * 32 * 32 * 2, 64);
  run<9>("  ... + barrier per 16 MFMAs", wgs, iters / 2, 2.0 * 32 * 32 * 2, 64);
  run<10>("  ... + barrier + LDS tile writes", wgs, iters / 2, 2.0 * 32 * 32 * 2, 64);
  run<12>("  ... + barrier + fetch via registers", wgs, iters / 2, 2.0 * 32 * 32 * 2, 64);
  run<11>("  ... + barrier + fetch via LDS-DMA", wgs, iters / 2, 2.0 * 32 * 32 * 2, 64);
  run<6>("v_mfma_f32_16x16x4_f32, random", wgs, iters / 2, 2.0 * 16 * 16 * 4, 32);
  run<7>("16x16x4 f32 + LDS operand reads", wgs, iters / 2, 2.0 * 16 * 16 * 4, 32);
  run<1>("v_mfma_f32_32x32x16_bf16", wgs, iters * 2, 2.0 * 32 * 32 * 16, 32);
  run<2>("v_fma chain (light) again", wgs, iters * 4, 0, 4);
  // a single workgroup: the clock without chip-wide load
  run<0>("f32 MFMA, ONE workgroup", 1, iters, 2.0 * 32 * 32 * 2, 64);
  return 0;
}
