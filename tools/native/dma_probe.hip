// L2 -> LDS delivery ceiling of a GEMM-shaped LDS-DMA stream on MI355X (no MFMA at all).
// Persistent workgroups walk the output tiles of C = A B^T (A: M x K, B: N x K, bf16, k contiguous) in the ring GEMM's order
// (XCD-contiguous chunks, groups of row panels walked column by column) and pull every 32-deep k-tile of both operand panels
// into an LDS ring with `buffer_load_dwordx4 ... lds` (16 rows x 64 bytes per instruction), counted vmcnt, one barrier per k-tile:
// exactly the load side of gemm_bf16_ring_kernel.  Optionally every wave also reads its MFMA fragments back with ds_read_b128.
// What it answers: how many bytes per clock and CU the memory pipeline delivers for this access pattern when nothing else limits
// the kernel, i.e. the ceiling the 256 x 256 (128 FLOP/B) and 256 x 128 (85 FLOP/B) tilings have to live under.
// Build: hipcc --offload-arch=gfx950 -O3 tools/native/dma_probe.hip -o tools/native/dma_probe
// Run:   tools/native/dma_probe [M=86680]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int xcd_chunk(int id, int n) {
  const int q = n >> 3, r = n & 7, xcd = id & 7, loc = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// READS: 0 none; 1: every wave reads (RA + RB) rows x 64 B of the slot as ds_read_b128 fragments (RA = BM / WM, RB = BN / WN)
template <int BM, int BN, int NW, int NS, int WM, int WN, int READS, int HOT>
__global__ void __launch_bounds__(NW * 64) probe(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int M, int N, int K, int ntiles,
                                                   int group_m, float* sink) {
  constexpr int SLOT = (BM + BN) * 64, D = NS - 1, PER = (BM + BN) / 16 / NW;
  static_assert((BM + BN) / 16 % NW == 0 && BM % 16 == 0, "shape");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  auto tile_mn = [&](int t, int& m0, int& n0) {
    const int per_group = group_m * tiles_n, grp_i = t / per_group, within = t - grp_i * per_group;
    const int rows = tiles_m - grp_i * group_m < group_m ? tiles_m - grp_i * group_m : group_m;
    m0 = (grp_i * group_m + within % rows) * BM;
    n0 = (within / rows) * BN;
  };
  const int srow = lane >> 2, sc = (lane & 3) ^ ((lane >> 4) & 3);
  const int nkt = (K + 31) / 32;
  int ltile = blockIdx.x, lt = 0, lslot = 0;
  __amdgpu_buffer_rsrc_t rsA, rsB;
  auto set_tile = [&](int v) {
    int m0, n0;
    tile_mn(xcd_chunk(v, ntiles), m0, n0);
    long long ab = ((long long)(M - 1 - m0) * K + K) * 2, bb = ((long long)(N - 1 - n0) * K + K) * 2;
    if (ab > 0x7FFFFFF0ll) ab = 0x7FFFFFF0ll;
    if (bb > 0x7FFFFFF0ll) bb = 0x7FFFFFF0ll;
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A + (long long)m0 * K), 0, (int)ab, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B + (long long)n0 * K), 0, (int)bb, 0x00020000);
  };
  set_tile(ltile);
  auto issue_next = [&]() {
    unsigned char* dst = smem + lslot * SLOT;
    const unsigned dead = ltile < ntiles ? 0u : 0x80000000u;
    const unsigned kt = HOT ? 0u : (unsigned)lt;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int rg = i * NW + wave;               // 16-row group of the (A | B) row list
      const int row = rg * 16 + srow;
      if (rg * 16 < BM) {
        const unsigned off = ((unsigned)row * (unsigned)K + sc * 8u) * 2u + kt * 64u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + rg * 1024), 16, off | dead, 0, 0, 0);
      } else {
        const unsigned off = ((unsigned)(row - BM) * (unsigned)K + sc * 8u) * 2u + kt * 64u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(dst + rg * 1024), 16, off | dead, 0, 0, 0);
      }
    }
    lslot = lslot + 1 == NS ? 0 : lslot + 1;
    if (++lt == nkt) {
      lt = 0;
      ltile += gridDim.x;
      if (ltile < ntiles) set_tile(ltile);
    }
  };
  const int li = lane & 15, g16 = lane >> 4;
  const int wr = wave / WN, wc = wave % WN;
  constexpr int RA = BM / WM / 16, RB = BN / WN / 16;   // 16-row fragment reads per wave and k-tile
  const unsigned a16 = (unsigned)(wr * (BM / WM) + li) * 64u + (unsigned)g16 * 16u;
  const unsigned b16 = (unsigned)(BM + wc * (BN / WN) + li) * 64u + (unsigned)g16 * 16u;
  float acc = 0.f;
  int ctile = blockIdx.x, ct = 0, cslot = 0;
#pragma unroll
  for (int t = 0; t < D; ++t) issue_next();
  wait_vmcnt<PER * (D - 1)>();
  __builtin_amdgcn_s_barrier();
  while (ctile < ntiles) {
    const unsigned char* sb = smem + cslot * SLOT;
    if constexpr (READS) {
      bf16x8 fa[RA], fb[RB];
#pragma unroll
      for (int i = 0; i < RA; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sb + a16 + i * 1024);
#pragma unroll
      for (int j = 0; j < RB; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(sb + b16 + j * 1024);
#pragma unroll
      for (int i = 0; i < RA; ++i) acc += (float)fa[i][0];
#pragma unroll
      for (int j = 0; j < RB; ++j) acc += (float)fb[j][1];
    }
    issue_next();
    wait_vmcnt<PER * (D - 1)>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    cslot = cslot + 1 == NS ? 0 : cslot + 1;
    if (++ct == nkt) {
      ct = 0;
      ctile += gridDim.x;
    }
  }
  wait_vmcnt<0>();
  if (acc == 1.2345e33f) sink[0] = acc;
}

template <int BM, int BN, int NW, int NS, int WM, int WN, int READS, int HOT>
void run(const char* name, const bf16_t* A, const bf16_t* B, int M, int N, int K, int wg_per_cu, float* sink) {
  constexpr int LDS = NS * (BM + BN) * 64;
  auto kern = probe<BM, BN, NW, NS, WM, WN, READS, HOT>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
    printf("%s: cannot set LDS %d\n", name, LDS);
    return;
  }
  const int ntiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int grid = 256 * wg_per_cu < ntiles ? 256 * wg_per_cu : ntiles;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS, 0, A, B, M, N, K, ntiles, 8, sink);
  hipEventRecord(e0, 0);
  const int iters = 10;
  for (int w = 0; w < iters; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS, 0, A, B, M, N, K, ntiles, 8, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  const double bytes = (double)ntiles * ((K + 31) / 32) * (BM + BN) * 64.0;
  const double flops = 2.0 * M * N * K;
  printf("%-34s M %6d N %5d K %5d  %2d WG/CU  %8.1f us  %7.2f TB/s = %6.1f GB/s/CU = %5.1f B/clk/CU @2.4GHz | a GEMM at this rate: %7.1f TFLOP/s  (err %s)\n",
         name, M, N, K, wg_per_cu, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4, flops / ms / 1e9,
         hipGetErrorString(hipGetLastError()));
}


// ---- same stream cut into 32 KB pieces whose rows are ROWB bytes (64: 32-deep k-tiles, 128: 64-deep, 256: 128-deep): does the memory pipeline
// deliver more when a DMA instruction covers whole 128-byte lines instead of 64-byte half lines?  (one instruction = 1 KB = 1024 / ROWB rows)
template <int ROWB, int NW, int NS, int HOT>
__global__ void __launch_bounds__(NW * 64) probe_rows(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int M, int N, int K, int ntiles,
                                                        int group_m, float* sink) {
  constexpr int BM = 256, BN = 256, PIECE = 32768, PR = PIECE / ROWB, D = NS - 1, PER = 32 / NW, RPI = 1024 / ROWB, LPR = ROWB / 16;
  constexpr int NPIECE = (BM + BN) / PR;      // pieces per k-tile
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  auto tile_mn = [&](int t, int& m0, int& n0) {
    const int per_group = group_m * tiles_n, grp_i = t / per_group, within = t - grp_i * per_group;
    const int rows = tiles_m - grp_i * group_m < group_m ? tiles_m - grp_i * group_m : group_m;
    m0 = (grp_i * group_m + within % rows) * BM;
    n0 = (within / rows) * BN;
  };
  const int srow = lane / LPR, sc = lane % LPR;
  const int nkt = (K * 2 + ROWB - 1) / ROWB;
  int ltile = blockIdx.x, lt = 0, lp = 0, lslot = 0;
  __amdgpu_buffer_rsrc_t rsA, rsB;
  auto set_tile = [&](int v) {
    int m0, n0;
    tile_mn(xcd_chunk(v, ntiles), m0, n0);
    long long ab = ((long long)(M - 1 - m0) * K + K) * 2, bb = ((long long)(N - 1 - n0) * K + K) * 2;
    if (ab > 0x7FFFFFF0ll) ab = 0x7FFFFFF0ll;
    if (bb > 0x7FFFFFF0ll) bb = 0x7FFFFFF0ll;
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A + (long long)m0 * K), 0, (int)ab, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B + (long long)n0 * K), 0, (int)bb, 0x00020000);
  };
  set_tile(ltile);
  auto issue_next = [&]() {
    unsigned char* dst = smem + lslot * PIECE;
    const unsigned dead = ltile < ntiles ? 0u : 0x80000000u;
    const unsigned kt = HOT ? 0u : (unsigned)lt;
    const int row0 = lp * PR;                    // first row of this piece in the (A | B) row list of the k-tile
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int row = row0 + (i * NW + wave) * RPI + srow;
      const unsigned off = ((unsigned)(row < BM ? row : row - BM) * (unsigned)K * 2u + sc * 16u + kt * (unsigned)ROWB) | dead;
      if (row0 < BM) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + (i * NW + wave) * 1024), 16, off, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(dst + (i * NW + wave) * 1024), 16, off, 0, 0, 0);
    }
    lslot = lslot + 1 == NS ? 0 : lslot + 1;
    if (++lp == NPIECE) {
      lp = 0;
      if (++lt == nkt) {
        lt = 0;
        ltile += gridDim.x;
        if (ltile < ntiles) set_tile(ltile);
      }
    }
  };
  int ctile = blockIdx.x, ct = 0;
#pragma unroll
  for (int t = 0; t < D; ++t) issue_next();
  wait_vmcnt<PER * (D - 1)>();
  __builtin_amdgcn_s_barrier();
  while (ctile < ntiles) {
    issue_next();
    wait_vmcnt<PER * (D - 1)>();
    __builtin_amdgcn_s_barrier();
    if (++ct == nkt * NPIECE) {
      ct = 0;
      ctile += gridDim.x;
    }
  }
  wait_vmcnt<0>();
}

template <int ROWB, int NW, int NS, int HOT>
void run_rows(const char* name, const bf16_t* A, const bf16_t* B, int M, int N, int K, float* sink) {
  constexpr int LDS = NS * 32768;
  auto kern = probe_rows<ROWB, NW, NS, HOT>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
    printf("%s: cannot set LDS %d\n", name, LDS);
    return;
  }
  const int ntiles = ((M + 255) / 256) * ((N + 255) / 256);
  const int grid = 256 < ntiles ? 256 : ntiles;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS, 0, A, B, M, N, K, ntiles, 8, sink);
  hipEventRecord(e0, 0);
  const int iters = 10;
  for (int w = 0; w < iters; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS, 0, A, B, M, N, K, ntiles, 8, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  const double bytes = (double)ntiles * K * 2.0 * 512.0;
  const double flops = 2.0 * M * N * K;
  printf("%-34s M %6d N %5d K %5d   1 WG/CU  %8.1f us  %7.2f TB/s = %6.1f GB/s/CU = %5.1f B/clk/CU @2.4GHz | a GEMM at this rate: %7.1f TFLOP/s  (err %s)\n",
         name, M, N, K, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4, flops / ms / 1e9,
         hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 86680;
  const int Kmax = 3072, Nmax = 3072;
  bf16_t *A, *B;
  float* sink;
  hipMalloc(&A, (size_t)M * Kmax * 2);
  hipMalloc(&B, (size_t)Nmax * Kmax * 2);
  hipMalloc(&sink, 16);
  hipMemset(A, 0x3c, (size_t)M * Kmax * 2);
  hipMemset(B, 0x3c, (size_t)Nmax * Kmax * 2);
  const int shapes[4][2] = {{2304, 768}, {768, 768}, {3072, 768}, {768, 3072}};
  for (int s = 0; s < 4; ++s) {
    const int N = shapes[s][0], K = shapes[s][1];
    printf("---- N %d K %d\n", N, K);
    run<256, 256, 8, 4, 2, 4, 0, 0>("256x256 8w ring4 dma only", A, B, M, N, K, 1, sink);
    run<256, 256, 8, 4, 2, 4, 1, 0>("256x256 8w ring4 dma+reads(128x64)", A, B, M, N, K, 1, sink);
    run<256, 256, 4, 4, 2, 2, 0, 0>("256x256 4w ring4 dma only", A, B, M, N, K, 1, sink);
    run<256, 256, 4, 4, 2, 2, 1, 0>("256x256 4w ring4 dma+reads(128x128)", A, B, M, N, K, 1, sink);
    run<256, 256, 8, 4, 2, 4, 0, 1>("256x256 8w ring4 dma only, HOT", A, B, M, N, K, 1, sink);
    run<256, 128, 4, 3, 2, 2, 0, 0>("256x128 4w ring3 dma only", A, B, M, N, K, 2, sink);
    run<256, 128, 4, 3, 2, 2, 1, 0>("256x128 4w ring3 dma+reads(128x64)", A, B, M, N, K, 2, sink);
    run<128, 256, 4, 3, 2, 2, 0, 0>("128x256 4w ring3 dma only", A, B, M, N, K, 2, sink);
    run<256, 128, 4, 3, 2, 2, 0, 1>("256x128 4w ring3 dma only, HOT", A, B, M, N, K, 2, sink);
    run_rows<64, 8, 4, 0>("256x256 8w 32KB pieces, 64-B rows", A, B, M, N, K, sink);
    run_rows<128, 8, 4, 0>("256x256 8w 32KB pieces, 128-B rows", A, B, M, N, K, sink);
    run_rows<256, 8, 4, 0>("256x256 8w 32KB pieces, 256-B rows", A, B, M, N, K, sink);
    run_rows<128, 4, 4, 0>("256x256 4w 32KB pieces, 128-B rows", A, B, M, N, K, sink);
    run_rows<128, 8, 5, 0>("256x256 8w 32KB x5, 128-B rows", A, B, M, N, K, sink);
    run_rows<128, 8, 3, 0>("256x256 8w 32KB x3, 128-B rows", A, B, M, N, K, sink);
    run_rows<128, 8, 4, 1>("256x256 8w 128-B rows, HOT", A, B, M, N, K, sink);
    run_rows<256, 8, 4, 1>("256x256 8w 256-B rows, HOT", A, B, M, N, K, sink);
    run<128, 128, 4, 4, 2, 2, 0, 0>("128x128 4w ring4 dma only", A, B, M, N, K, 2, sink);
    run<128, 128, 4, 3, 2, 2, 0, 0>("128x128 4w ring3 dma only", A, B, M, N, K, 3, sink);
  }
  hipDeviceSynchronize();
  return 0;
}
