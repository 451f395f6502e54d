// Memory-pattern probe for tg_expand_i8 at S=4 (B parents, k=8 children each, no arithmetic): what the write stream
// costs with plain and with non-temporal stores, with and without the token / done / changed traffic.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result tools/expand_probe.hip -o /tmp/ep && /tmp/ep 1048576
// Output of one run: profiles/r02_expand_probe.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int V>
__global__ __launch_bounds__(256) void k(const uint4* in, uint4* out, const int* tok, uint8_t* done, uint8_t* chg, long long nchild) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long child = t >> 2;
  if (child >= nchild) return;
  const int q = t & 3;
  uint4 p = in[(child >> 3) * 4 + q];
  if (V >= 1) {
    const int a = tok[child * 3], b = tok[child * 3 + 1], c = tok[child * 3 + 2];
    p.x ^= (a ^ b ^ c) == 0x12345678;
  }
  if (V == 4) __builtin_nontemporal_store(*reinterpret_cast<v4u*>(&p), reinterpret_cast<v4u*>(out) + t);
  else out[t] = p;
  if (V >= 2 && q == 0) {
    done[child] = (p.x | p.y | p.z | p.w) == 0;
    if (V >= 3) chg[child] = p.x != 7;
  }
}
// per-parent team: 4 lanes write the 8 children in a loop (64-byte pieces 512 bytes apart per team... contiguous per parent)
__global__ __launch_bounds__(256) void kp(const uint4* in, uint4* out, const int* tok, uint8_t* done, uint8_t* chg, long long B) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long par = t >> 2;
  if (par >= B) return;
  const int q = t & 3;
  const uint4 p = in[t];
  for (int c = 0; c < 8; ++c) out[(par * 8 + c) * 4 + q] = p;
}
// a wavefront takes 2 parents; lane l: child = l>>2 -> same as k<0>; variant: thread per 16 B, 4 chunks per thread strided by wave (more bytes in flight per wave)
__global__ __launch_bounds__(256) void k4(const uint4* in, uint4* out, long long nchild) {
  const long long t0 = ((long long)blockIdx.x * 256 + (threadIdx.x & ~63)) * 4 + (threadIdx.x & 63);
  uint4 p[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const long long t = t0 + 64 * u; p[u] = in[((t >> 2) >> 3) * 4 + (t & 3)]; }
#pragma unroll
  for (int u = 0; u < 4; ++u) { const long long t = t0 + 64 * u; if ((t >> 2) < nchild) out[t] = p[u]; }
}
__global__ __launch_bounds__(256) void kfill(uint4* out, long long n16) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < n16) out[t] = uint4{1, 2, 3, 4};
}
int main(int argc, char** argv) {
  const long long B = argc > 1 ? atoll(argv[1]) : (1 << 20);
  const long long nchild = B * 8;
  uint4 *in, *out; int* tok; uint8_t *done, *chg;
  CK(hipMalloc(&in, B * 64)); CK(hipMalloc(&out, nchild * 64)); CK(hipMalloc(&tok, nchild * 12));
  CK(hipMalloc(&done, nchild)); CK(hipMalloc(&chg, nchild));
  CK(hipMemset(in, 1, B * 64)); CK(hipMemset(tok, 1, nchild * 12));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((nchild * 4 + 255) / 256);
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < 10; ++i) launch();
    (void)hipEventRecord(e1, s);
    (void)hipStreamSynchronize(s);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f us  %6.0f GB/s\n", name, ms * 100, bytes / (ms * 1e-4) / 1e9);
    return 0;
  };
  const double wb = nchild * 64.0, rb = B * 64.0, tb = nchild * 12.0;
  run("fill (16 B per thread)", [&] { hipLaunchKernelGGL(kfill, dim3(grid), dim3(256), 0, s, out, nchild * 4); }, wb);
  run("child copy", [&] { hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, s, in, out, tok, done, chg, nchild); }, wb + rb);
  run("child copy + tokens", [&] { hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, s, in, out, tok, done, chg, nchild); }, wb + rb + tb);
  run("child copy + tokens + done", [&] { hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, s, in, out, tok, done, chg, nchild); }, wb + rb + tb + nchild);
  run("child copy + tokens + done + changed", [&] { hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, s, in, out, tok, done, chg, nchild); }, wb + rb + tb + 2 * nchild);
  run("same, nontemporal child stores", [&] { hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, s, in, out, tok, done, chg, nchild); }, wb + rb + tb + 2 * nchild);
  run("team per parent, loop over children", [&] { hipLaunchKernelGGL(kp, dim3((unsigned)((B * 4 + 255) / 256)), dim3(256), 0, s, in, out, tok, done, chg, B); }, wb + rb);
  run("child copy, 4 chunks per thread", [&] { hipLaunchKernelGGL(k4, dim3(grid / 4), dim3(256), 0, s, in, out, nchild); }, wb + rb);
  return 0;
}
