// How fast can one MI355X READ?  The in-place S >= 9 step kernels skip the stores of unchanged rows, so they are
// read streams (537 MB at S=16 B=131072 in 148 us = 3.6 TB/s) -- is that the chip's read-only rate or the kernels'?
// Reads N bytes with U independent 16-byte loads per thread per trip (grid-stride or one trip), XOR-folds them, stores
// one dword per thread only if the fold hits a magic value (never).  Also a copy (read + write) of the same bytes.
// Build: hipcc -O3 --offload-arch=gfx950 tools/read_bw_probe.hip -o read_bw_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ p, size_t n16, unsigned* sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * U) {
    uint4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      const uint4* a = p + (j < n16 ? j : i);
      if (NT) {
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        const u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(a));
        q[u] = uint4{t[0], t[1], t[2], t[3]};
      } else {
        q[u] = *a;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x ^= q[u].x; acc.y ^= q[u].y; acc.z ^= q[u].z; acc.w ^= q[u].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[threadIdx.x] = acc.x;
}

// one game-sized block (bytes_per_wg) per workgroup, all loads issued at once: the shape of the step kernels
template <int U>
__global__ __launch_bounds__(256) void k_read_block(const uint4* __restrict__ p, size_t n16, unsigned* sink) {
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  uint4 q[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const size_t j = base + 256 * u; q[u] = p[j < n16 ? j : 0]; }
  uint4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < U; ++u) { acc.x ^= q[u].x; acc.y ^= q[u].y; acc.z ^= q[u].z; acc.w ^= q[u].w; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[threadIdx.x] = acc.x;
}

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ p, uint4* __restrict__ o, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) o[i] = p[i];
}

int main() {
  const size_t sizes[] = {64ull << 20, 512ull << 20};
  uint4 *a, *b; unsigned* sink;
  CK(hipMalloc(&a, 512ull << 20)); CK(hipMalloc(&b, 512ull << 20)); CK(hipMalloc(&sink, 4096));
  CK(hipMemset(a, 1, 512ull << 20)); CK(hipMemset(b, 2, 512ull << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  auto time = [&](auto launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best * 1e-3;
  };
  for (size_t bytes : sizes) {
    const size_t n16 = bytes / 16;
    printf("---- %zu MiB ----\n", bytes >> 20);
    for (int wgs_per_cu : {4, 8}) {
      const int grid = cus * wgs_per_cu;
      double s;
      s = time([&] { hipLaunchKernelGGL((k_read<1, false>), dim3(grid), dim3(256), 0, 0, a, n16, sink); });
      printf("read, grid-stride, %d WG/CU, U=1           : %7.1f us  %6.0f GB/s\n", wgs_per_cu, s * 1e6, bytes / s / 1e9);
      s = time([&] { hipLaunchKernelGGL((k_read<4, false>), dim3(grid), dim3(256), 0, 0, a, n16, sink); });
      printf("read, grid-stride, %d WG/CU, U=4           : %7.1f us  %6.0f GB/s\n", wgs_per_cu, s * 1e6, bytes / s / 1e9);
      s = time([&] { hipLaunchKernelGGL((k_read<8, false>), dim3(grid), dim3(256), 0, 0, a, n16, sink); });
      printf("read, grid-stride, %d WG/CU, U=8           : %7.1f us  %6.0f GB/s\n", wgs_per_cu, s * 1e6, bytes / s / 1e9);
      s = time([&] { hipLaunchKernelGGL((k_read<8, true>), dim3(grid), dim3(256), 0, 0, a, n16, sink); });
      printf("read, grid-stride, %d WG/CU, U=8, nt loads : %7.1f us  %6.0f GB/s\n", wgs_per_cu, s * 1e6, bytes / s / 1e9);
    }
    {
      double s = time([&] { hipLaunchKernelGGL((k_read_block<4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, a, n16, sink); });
      printf("read, one 16 KiB block per WG (U=4)        : %7.1f us  %6.0f GB/s\n", s * 1e6, bytes / s / 1e9);
      s = time([&] { hipLaunchKernelGGL((k_read_block<1>), dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, a, n16, sink); });
      printf("read, one 4 KiB block per WG (U=1)         : %7.1f us  %6.0f GB/s\n", s * 1e6, bytes / s / 1e9);
      s = time([&] { hipLaunchKernelGGL(k_copy, dim3(cus * 8), dim3(256), 0, 0, a, b, n16); });
      printf("copy (read + write), 8 WG/CU               : %7.1f us  %6.0f GB/s moved\n", s * 1e6, 2.0 * bytes / s / 1e9);
    }
  }
  return 0;
}
