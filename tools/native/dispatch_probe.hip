// Workgroup dispatch throughput: how fast can the chip START workgroups of the fp32 GEMM's shape (256 threads, ~30 KB LDS, ~96
// VGPRs) when each lives `work` shader cycles?  If G workgroups x `work` cycles over 256 CUs x 5 slots finish later than the
// slot-limited time, the difference is dispatch (launch gaps), not the kernel body.
// Build: hipcc --offload-arch=gfx950 -O3 tools/native/dispatch_probe.hip -o tools/native/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) body(float* sink, int work, int touch) {
  extern __shared__ float lds[];
  const long long t0 = __builtin_readcyclecounter();
  if (touch) lds[threadIdx.x] = (float)threadIdx.x;      // make the allocation real
  while (__builtin_readcyclecounter() - t0 < work) __builtin_amdgcn_s_sleep(8);
  if (touch && lds[(threadIdx.x + 1) & 255] == -1.f) sink[0] = 1.f;
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* sink;
  (void)hipMalloc(&sink, 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(body), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int grids[] = {1280, 4800, 6400, 25600};
  const int works[] = {0, 5000, 20000, 70000};
  const int ldss[] = {0, 30720, 65536};
  printf("%d CUs. columns: grid, LDS bytes (slots per CU), work cycles -> measured us | slot-limited us at 2.1 GHz | workgroups started per us\n", cus);
  for (int lds : ldss) {
    const int slots = lds ? (160 * 1024) / lds : 8;
    for (int g : grids)
      for (int w : works) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          (void)hipEventRecord(e0);
          hipLaunchKernelGGL(body, dim3(g), dim3(256), lds, 0, sink, w, lds ? 1 : 0);
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
          float ms;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        const double rounds = (double)((g + cus * slots - 1) / (cus * slots));
        printf("grid %6d  lds %6d (%d/CU)  work %6d : %8.1f us | %8.1f us | %6.1f wg/us\n", g, lds, slots > 8 ? 8 : slots, w, best * 1e3, rounds * w / 2100.0,
               g / (best * 1e3));
      }
  }
  return 0;
}
