// Memory-pattern probe for tg_expand_i8 at S=25 (4096 parents, k = 8 children of 15 625 bytes on a 15 632-byte stride, no
// arithmetic): what does the write stream of "one workgroup per parent, its four chunks per thread stored k times" cost,
// with plain and non-temporal stores -- i.e. what could a compacted S=25 expand reach at best?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result tools/expand25_probe.hip -o /tmp/e25 && /tmp/e25
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr int NCH = 977, STRIDE = 15632, TSA = 250;  // chunks per game (the last holds 9 bytes), threads that own chunks
template <bool NT, bool BAR>
__global__ __launch_bounds__(256) void kparent(const uint8_t* in, uint8_t* out, int k) {
  const long long g = blockIdx.x;
  const int lt = threadIdx.x;
  uint4 p[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int c = lt + TSA * n;
    p[n] = (lt < TSA && c < NCH) ? *reinterpret_cast<const uint4*>(in + g * STRIDE + 16 * (c < NCH - 1 ? c : NCH - 2)) : uint4{0, 0, 0, 0};
  }
  for (int ch = 0; ch < k; ++ch) {
    uint8_t* o = out + (g * k + ch) * STRIDE;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int c = lt + TSA * n;
      if (lt < TSA && c < NCH - 1) {
        uint4 q = p[n];
        q.x += ch;
        if (NT) __builtin_nontemporal_store(*reinterpret_cast<v4u*>(&q), reinterpret_cast<v4u*>(o + 16 * c));
        else *reinterpret_cast<uint4*>(o + 16 * c) = q;
      }
    }
    if (BAR) __syncthreads();
  }
}
// one workgroup per CHILD (the parent is read k times, from the caches): consecutive workgroups write consecutive memory
template <bool NT>
__global__ __launch_bounds__(256) void kchild(const uint8_t* in, uint8_t* out, int k) {
  const long long g = blockIdx.x, par = g / k;
  const int lt = threadIdx.x;
  uint8_t* o = out + g * STRIDE;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int c = lt + TSA * n;
    if (lt < TSA && c < NCH - 1) {
      uint4 q = *reinterpret_cast<const uint4*>(in + par * STRIDE + 16 * c);
      q.x += static_cast<uint32_t>(g);
      if (NT) __builtin_nontemporal_store(*reinterpret_cast<v4u*>(&q), reinterpret_cast<v4u*>(o + 16 * c));
      else *reinterpret_cast<uint4*>(o + 16 * c) = q;
    }
  }
}
// one workgroup per parent, but the children of a parent are written INTERLEAVED with the other parents' by ordering the
// loop child-major across the grid: workgroup g writes child (ch) of parent g in trip ch -- as kparent -- only the OUTPUT is
// laid out child-major (B x k -> k x B): is it the k-strided windows that cost?
template <bool NT>
__global__ __launch_bounds__(256) void kparent_cm(const uint8_t* in, uint8_t* out, int k, long long B) {
  const long long g = blockIdx.x;
  const int lt = threadIdx.x;
  uint4 p[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int c = lt + TSA * n;
    p[n] = (lt < TSA && c < NCH - 1) ? *reinterpret_cast<const uint4*>(in + g * STRIDE + 16 * c) : uint4{0, 0, 0, 0};
  }
  for (int ch = 0; ch < k; ++ch) {
    uint8_t* o = out + (ch * B + g) * STRIDE;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int c = lt + TSA * n;
      if (lt < TSA && c < NCH - 1) {
        uint4 q = p[n];
        q.x += ch;
        if (NT) __builtin_nontemporal_store(*reinterpret_cast<v4u*>(&q), reinterpret_cast<v4u*>(o + 16 * c));
        else *reinterpret_cast<uint4*>(o + 16 * c) = q;
      }
    }
  }
}
__global__ __launch_bounds__(256) void kfill(uint4* out, long long n16) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < n16) out[t] = uint4{1, 2, 3, 4};
}
int main(int argc, char** argv) {
  const long long B = argc > 1 ? atoll(argv[1]) : 4096;
  const int k = 8;
  uint8_t *in, *out;
  CK(hipMalloc(&in, B * STRIDE)); CK(hipMalloc(&out, B * k * STRIDE));
  CK(hipMemset(in, 1, B * STRIDE));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 200; ++i) launch();
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e1, s);
    (void)hipStreamSynchronize(s);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.1f us  %6.0f GB/s\n", name, ms * 50, bytes / (ms * 0.5e-4) / 1e9);
    return 0;
  };
  const double wb = (double)B * k * 15625, rb = (double)B * 15625;
  run("fill of the children (16 B per thread)", [&] { hipLaunchKernelGGL(kfill, dim3((unsigned)((B * k * STRIDE / 16 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint4*>(out), B * k * STRIDE / 16); }, wb);
  run("workgroup per parent, k children, plain stores", [&] { hipLaunchKernelGGL((kparent<false, false>), dim3((unsigned)B), dim3(256), 0, s, in, out, k); }, wb + rb);
  run("workgroup per parent, k children, nt stores", [&] { hipLaunchKernelGGL((kparent<true, false>), dim3((unsigned)B), dim3(256), 0, s, in, out, k); }, wb + rb);
  run("... plain stores, a barrier per child", [&] { hipLaunchKernelGGL((kparent<false, true>), dim3((unsigned)B), dim3(256), 0, s, in, out, k); }, wb + rb);
  run("... nt stores, a barrier per child", [&] { hipLaunchKernelGGL((kparent<true, true>), dim3((unsigned)B), dim3(256), 0, s, in, out, k); }, wb + rb);
  for (int pad : {16384, 32768, 49152, 65536}) {  // fewer resident workgroups per CU (unused dynamic LDS)
    char name[96];
    snprintf(name, sizeof name, "workgroup per parent, plain stores, %d KB of unused LDS", pad >> 10);
    run(name, [&] { hipLaunchKernelGGL((kparent<false, false>), dim3((unsigned)B), dim3(256), pad, s, in, out, k); }, wb + rb);
  }
  run("workgroup per CHILD (parent re-read), plain stores", [&] { hipLaunchKernelGGL((kchild<false>), dim3((unsigned)(B * k)), dim3(256), 0, s, in, out, k); }, wb + rb);
  run("workgroup per CHILD (parent re-read), nt stores", [&] { hipLaunchKernelGGL((kchild<true>), dim3((unsigned)(B * k)), dim3(256), 0, s, in, out, k); }, wb + rb);
  run("workgroup per parent, children laid out child-major, plain", [&] { hipLaunchKernelGGL((kparent_cm<false>), dim3((unsigned)B), dim3(256), 0, s, in, out, k, B); }, wb + rb);
  run("workgroup per parent, children laid out child-major, nt", [&] { hipLaunchKernelGGL((kparent_cm<true>), dim3((unsigned)B), dim3(256), 0, s, in, out, k, B); }, wb + rb);
  return 0;
}
