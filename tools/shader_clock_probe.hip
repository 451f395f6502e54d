// What does s_memtime count, and how fast does a CU issue VALU work under different loads?
//  (1) one wavefront spins for 20 M s_memtime ticks; events give the wall time -> tick rate;
//  (2) a dependent chain of 1 M v_add_u32 in one wavefront (known: 4 cycles issue + dependent latency) timed by events and by ticks;
//  (3) the same chain on every SIMD of the chip with 1, 4, 8 wavefronts per SIMD (events) -> the sustained VALU clock.
// Build: hipcc -O3 --offload-arch=gfx950 tools/shader_clock_probe.hip -o shader_clock_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_ticks(unsigned long long n, unsigned long long* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long t = t0;
  unsigned iters = 0;
  while (t - t0 < n && iters < (1u << 28)) { t = __builtin_amdgcn_s_memtime(); ++iters; }
  if (threadIdx.x == 0) out[0] = t - t0;
}

__global__ void chain(int n, unsigned* sink, unsigned long long* ticks) {
  unsigned x = threadIdx.x, y = blockIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (x == 0xdeadbeef) sink[0] = x;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  unsigned long long* d; unsigned* sink;
  CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms; unsigned long long h;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(spin_ticks, dim3(1), dim3(64), 0, 0, 20000000ull, d); CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("spin: %llu ticks in %.3f ms -> %.1f MHz\n", h, ms, h / (ms * 1e3));
  }
  const int n = 1 << 20;
  int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  struct { int blocks, threads; const char* what; } cfg[] = {{1, 64, "1 wavefront"}, {cus, 256, "1 wavefront per SIMD, whole chip"},
      {cus * 4, 256, "4 wavefronts per SIMD, whole chip"}, {cus * 8, 256, "8 wavefronts per SIMD, whole chip"}};
  for (auto& c : cfg) for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(chain, dim3(c.blocks), dim3(c.threads), 0, 0, n, sink, d); CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    const double wps = (double)c.blocks * c.threads / 64 / (cus * 4.0);  // wavefronts per SIMD
    printf("%-36s: %d v_add in %.3f ms; wave 0: %llu ticks (%.2f ticks per v_add); issue rate per SIMD %.3f G wave-instr/s\n", c.what, n, ms, h,
           (double)h / n, (wps < 1 ? 1 : wps) * n / (ms * 1e-3) / 1e9);
  }
  return 0;
}
