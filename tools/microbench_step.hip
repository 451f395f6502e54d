// Standalone microbenchmark (no torch): what bounds ONE tg_step launch at the BASELINE cfg2 shape
// (S=4, 65 536 games = 4 MiB of state)?  Measures, as hipGraph replays of N chained launches:
//   empty      -- a kernel with the same grid that does nothing (launch/boundary floor)
//   copy       -- 16 B load + 16 B store per lane + done byte (memory path floor)
//   variants of the step kernel (grid shape, games per lane, arithmetic form)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench_step.hip -o tools/microbench_step
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/tensor_game.h"

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

__device__ __forceinline__ int sbyte(uint32_t w, int t) { return __builtin_amdgcn_sbfe((int)w, 8 * t, 8); }
__device__ __forceinline__ uint32_t pack4(int n0, int n1, int n2, int n3) {
  uint32_t lo = __builtin_amdgcn_perm((uint32_t)n1, (uint32_t)n0, 0x0c0c0400u);
  uint32_t hi = __builtin_amdgcn_perm((uint32_t)n3, (uint32_t)n2, 0x0c0c0400u);
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_empty(const uint4* in, uint4* out, const int* tok, uint8_t* done, int B) {}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy(const uint4* in, uint4* out, const int* tok, uint8_t* done, int B) {
  const int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t < 4 * B) {
    uint4 q = in[t];
    out[t] = q;
    if ((t & 3) == 0) done[t >> 2] = (q.x | q.y | q.z | q.w) == 0;
  }
}

// reference form: 32-bit per element, as in tg_kernels.hip s4_kernel<STEP> but 32-bit indexing
template <int BLOCK, int GPL>  // GPL = games handled per 4-lane team (chunks per lane)
__global__ __launch_bounds__(BLOCK) void k_step32(const uint4* in, uint4* out, const int* tok, uint8_t* done, int B) {
  const int t0 = (blockIdx.x * BLOCK + threadIdx.x);
  const int q = t0 & 3;
  const int team = t0 >> 2;
  const int nteams = (gridDim.x * BLOCK) >> 2;
  uint4 pk[GPL];
  int tk[GPL][3];
#pragma unroll
  for (int n = 0; n < GPL; ++n) {
    int g = team + n * nteams;
    if (g >= B) g = B - 1;
    pk[n] = in[g * 4 + q];
    tk[n][0] = tok[g * 3];
    tk[n][1] = tok[g * 3 + 1];
    tk[n][2] = tok[g * 3 + 2];
  }
#pragma unroll
  for (int n = 0; n < GPL; ++n) {
    const int g = team + n * nteams;
    const int ui = -(__builtin_amdgcn_sbfe(tk[n][0], 8 * q, 8) - 1);
    int v[4], w[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      v[t] = sbyte(tk[n][1], t) - 1;
      w[t] = sbyte(tk[n][2], t) - 1;
    }
    const uint32_t wd[4] = {pk[n].x, pk[n].y, pk[n].z, pk[n].w};
    uint32_t o[4], nz = 0;
    int ovf = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int uv = __mul24(ui, v[j]);
      int r[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        r[l] = __mul24(uv, w[l]) + sbyte(wd[j], l);
        ovf |= r[l] + 128;
      }
      o[j] = pack4(r[0], r[1], r[2], r[3]);
      nz |= o[j];
    }
    const uint64_t m = __ballot(nz != 0);
    const int lane = threadIdx.x & 63;
    const bool any = ((m >> (lane & ~3)) & 0xf) != 0;
    if (g < B) {
      out[g * 4 + q] = uint4{o[0], o[1], o[2], o[3]};
      if (q == 0) done[g] = !any;
    }
    if (ovf & ~255) done[0] = 2;  // stand-in for the rare overflow store
  }
}

// biased form: state bytes ^ 0x80 are unsigned, n+128 is computed directly so the range check is one OR
template <int BLOCK, int GPL>
__global__ __launch_bounds__(BLOCK) void k_stepb(const uint4* in, uint4* out, const int* tok, uint8_t* done, int B) {
  const int t0 = (blockIdx.x * BLOCK + threadIdx.x);
  const int q = t0 & 3;
  const int team = t0 >> 2;
  const int nteams = (gridDim.x * BLOCK) >> 2;
  uint4 pk[GPL];
  int tk[GPL][3];
#pragma unroll
  for (int n = 0; n < GPL; ++n) {
    int g = team + n * nteams;
    if (g >= B) g = B - 1;
    pk[n] = in[g * 4 + q];
    tk[n][0] = tok[g * 3];
    tk[n][1] = tok[g * 3 + 1];
    tk[n][2] = tok[g * 3 + 2];
  }
#pragma unroll
  for (int n = 0; n < GPL; ++n) {
    const int g = team + n * nteams;
    const int ui = -(__builtin_amdgcn_sbfe(tk[n][0], 8 * q, 8) - 1);
    int v[4], w[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      v[t] = sbyte(tk[n][1], t) - 1;
      w[t] = sbyte(tk[n][2], t) - 1;
    }
    const uint32_t wd[4] = {pk[n].x ^ 0x80808080u, pk[n].y ^ 0x80808080u, pk[n].z ^ 0x80808080u,
                            pk[n].w ^ 0x80808080u};
    uint32_t o[4], nz = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int uv = __mul24(ui, v[j]);
      int r[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        r[l] = __mul24(uv, w[l]) + (int)__builtin_amdgcn_ubfe(wd[j], 8 * l, 8);
        ovf |= (uint32_t)r[l];
      }
      o[j] = pack4(r[0], r[1], r[2], r[3]) ^ 0x80808080u;
      nz |= o[j];
    }
    const uint64_t m = __ballot(nz != 0);
    const int lane = threadIdx.x & 63;
    const bool any = ((m >> (lane & ~3)) & 0xf) != 0;
    if (g < B) {
      out[g * 4 + q] = uint4{o[0], o[1], o[2], o[3]};
      if (q == 0) done[g] = !any;
    }
    if (ovf & ~255u) done[0] = 2;
  }
}

typedef unsigned v4u __attribute__((ext_vector_type(4)));
// copy with non-temporal stores / loads: does the end-of-kernel L2 write-back shrink when stores stream?
template <int BLOCK, int NTL>
__global__ __launch_bounds__(BLOCK) void k_copy_nt(const uint4* in, uint4* out, const int* tok, uint8_t* done, int B) {
  const int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t < 4 * B) {
    const v4u* src = reinterpret_cast<const v4u*>(in) + t;
    v4u q = NTL ? __builtin_nontemporal_load(src) : *src;
    __builtin_nontemporal_store(q, reinterpret_cast<v4u*>(out) + t);
    if ((t & 3) == 0) done[t >> 2] = (q.x | q.y | q.z | q.w) == 0;
  }
}

// copy + the three token dwords (no arithmetic): what the memory side of a step costs
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copytok(const uint4* in, uint4* out, const int* tok, uint8_t* done, int B) {
  const int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t < 4 * B) {
    uint4 q = in[t];
    const int g = t >> 2;
    const int a = tok[g * 3], b = tok[g * 3 + 1], c = tok[g * 3 + 2];
    out[t] = q;
    if ((t & 3) == 0) done[g] = ((q.x | q.y | q.z | q.w) == 0) | ((a ^ b ^ c) == 0x7fffffff);
  }
}

// packed int16 form: 8 saturating v_pk_mad_i16 on sign-extended pairs; no factor range checks
// (saturation turns every inexact case into an int8 overflow, which is re-done in 32-bit)
__device__ __forceinline__ uint32_t pkmad_lo(uint32_t a, uint32_t b, uint32_t c) {  // a.lo broadcast
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t pkmad_hi(uint32_t a, uint32_t b, uint32_t c) {  // a.hi broadcast
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t pkmul_sat(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, 0 clamp" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t pksub(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t pkaddu(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_steppk(const uint4* in, uint4* out, const int* tok, uint8_t* done, int B) {
  const int t0 = (blockIdx.x * BLOCK + threadIdx.x);
  const int q = t0 & 3;
  int g = t0 >> 2;
  const bool live = g < B;
  if (!live) g = B - 1;
  const uint4 pk = in[g * 4 + q];
  const uint32_t du = tok[g * 3], dv = tok[g * 3 + 1], dw = tok[g * 3 + 2];
  const uint32_t shp = 0x00010001u;
  const int ui = 1 - __builtin_amdgcn_sbfe((int)du, 8 * q, 8);
  const uint32_t uip = __builtin_amdgcn_perm((uint32_t)ui, (uint32_t)ui, 0x05040100u);
  const uint32_t yv = dv << 8, yw = dw << 8;
  const uint32_t vA = pksub(__builtin_amdgcn_perm(dv, yv, 0x0A050804u), shp);
  const uint32_t vB = pksub(__builtin_amdgcn_perm(dv, yv, 0x0B070906u), shp);
  const uint32_t wA = pksub(__builtin_amdgcn_perm(dw, yw, 0x0A050804u), shp);
  const uint32_t wB = pksub(__builtin_amdgcn_perm(dw, yw, 0x0B070906u), shp);
  const uint32_t uvA = pkmul_sat(vA, uip), uvB = pkmul_sat(vB, uip);
  const uint32_t x[4] = {pk.x, pk.y, pk.z, pk.w};
  uint32_t A[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t y = x[d] << 8;
    const uint32_t lo = __builtin_amdgcn_perm(x[d], y, 0x0A050804u), hi = __builtin_amdgcn_perm(x[d], y, 0x0B070906u);
    const uint32_t uv = d < 2 ? uvA : uvB;
    if (d & 1) {
      A[2 * d] = pkmad_hi(uv, wA, lo);
      A[2 * d + 1] = pkmad_hi(uv, wB, hi);
    } else {
      A[2 * d] = pkmad_lo(uv, wA, lo);
      A[2 * d + 1] = pkmad_lo(uv, wB, hi);
    }
  }
  uint32_t o[4], nz = 0, ovf = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    ovf |= pkaddu(A[2 * d], 0x00800080u) | pkaddu(A[2 * d + 1], 0x00800080u);
    o[d] = __builtin_amdgcn_perm(A[2 * d + 1], A[2 * d], 0x06040200u);
    nz |= o[d];
  }
  if (ovf & 0xFF00FF00u) {  // rare: exact 32-bit redo of this lane's slice
    const int v[4] = {sbyte(dv, 0) - 1, sbyte(dv, 1) - 1, sbyte(dv, 2) - 1, sbyte(dv, 3) - 1};
    const int w[4] = {sbyte(dw, 0) - 1, sbyte(dw, 1) - 1, sbyte(dw, 2) - 1, sbyte(dw, 3) - 1};
    nz = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int uv = ui * v[j];
      o[j] = pack4(uv * w[0] + sbyte(x[j], 0), uv * w[1] + sbyte(x[j], 1), uv * w[2] + sbyte(x[j], 2),
                   uv * w[3] + sbyte(x[j], 3));
      nz |= o[j];
    }
    done[0] = 2;  // stand-in for the overflow store
  }
  const uint64_t m = __ballot(nz != 0);
  const int lane = threadIdx.x & 63;
  const bool any = ((m >> (lane & ~3)) & 0xf) != 0;
  if (live) {
    out[g * 4 + q] = uint4{o[0], o[1], o[2], o[3]};
    if (q == 0) done[g] = !any;
  }
}

struct FatArgs {  // the product's ApplyArgs layout: 96 bytes, the fields a step needs spread over it
  const uint4* in; uint4* out; const int* tok; uint8_t* done; int* done_step; uint8_t* changed; uint8_t* overflow;
  long long B, in_stride, out_stride; int S, nact, shift;
};
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_steppk_fat(FatArgs a) {
  const uint4* in = a.in; uint4* out = a.out; const int* tok = a.tok; uint8_t* done = a.done; const int B = (int)a.B;
  const int t0 = (blockIdx.x * BLOCK + threadIdx.x);
  const int q = t0 & 3;
  int g = t0 >> 2;
  const bool live = g < B;
  if (!live) g = B - 1;
  const uint4 pk = in[g * 4 + q];
  const uint32_t du = tok[g * 3], dv = tok[g * 3 + 1], dw = tok[g * 3 + 2];
  const uint32_t shp = ((uint32_t)a.shift & 0xffffu) | ((uint32_t)a.shift << 16);
  const int ui = a.shift - __builtin_amdgcn_sbfe((int)du, 8 * q, 8);
  const uint32_t uip = __builtin_amdgcn_perm((uint32_t)ui, (uint32_t)ui, 0x05040100u);
  const uint32_t yv = dv << 8, yw = dw << 8;
  const uint32_t vA = pksub(__builtin_amdgcn_perm(dv, yv, 0x0A050804u), shp);
  const uint32_t vB = pksub(__builtin_amdgcn_perm(dv, yv, 0x0B070906u), shp);
  const uint32_t wA = pksub(__builtin_amdgcn_perm(dw, yw, 0x0A050804u), shp);
  const uint32_t wB = pksub(__builtin_amdgcn_perm(dw, yw, 0x0B070906u), shp);
  const uint32_t uvA = pkmul_sat(vA, uip), uvB = pkmul_sat(vB, uip);
  const uint32_t x[4] = {pk.x, pk.y, pk.z, pk.w};
  uint32_t A[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t y = x[d] << 8;
    const uint32_t lo = __builtin_amdgcn_perm(x[d], y, 0x0A050804u), hi = __builtin_amdgcn_perm(x[d], y, 0x0B070906u);
    const uint32_t uv = d < 2 ? uvA : uvB;
    if (d & 1) {
      A[2 * d] = pkmad_hi(uv, wA, lo);
      A[2 * d + 1] = pkmad_hi(uv, wB, hi);
    } else {
      A[2 * d] = pkmad_lo(uv, wA, lo);
      A[2 * d + 1] = pkmad_lo(uv, wB, hi);
    }
  }
  uint32_t o[4], nz = 0, ovf = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    ovf |= pkaddu(A[2 * d], 0x00800080u) | pkaddu(A[2 * d + 1], 0x00800080u);
    o[d] = __builtin_amdgcn_perm(A[2 * d + 1], A[2 * d], 0x06040200u);
    nz |= o[d];
  }
  if (ovf & 0xFF00FF00u) {  // rare: exact 32-bit redo of this lane's slice
    const int v[4] = {sbyte(dv, 0) - 1, sbyte(dv, 1) - 1, sbyte(dv, 2) - 1, sbyte(dv, 3) - 1};
    const int w[4] = {sbyte(dw, 0) - 1, sbyte(dw, 1) - 1, sbyte(dw, 2) - 1, sbyte(dw, 3) - 1};
    nz = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int uv = ui * v[j];
      o[j] = pack4(uv * w[0] + sbyte(x[j], 0), uv * w[1] + sbyte(x[j], 1), uv * w[2] + sbyte(x[j], 2),
                   uv * w[3] + sbyte(x[j], 3));
      nz |= o[j];
    }
    done[0] = 2;  // stand-in for the overflow store
  }
  const uint64_t m = __ballot(nz != 0);
  const int lane = threadIdx.x & 63;
  const bool any = ((m >> (lane & ~3)) & 0xf) != 0;
  if (live) {
    out[g * 4 + q] = uint4{o[0], o[1], o[2], o[3]};
    if (q == 0) done[g] = !any;
  }
}

struct Variant {
  const char* name;
  void (*kern)(const uint4*, uint4*, const int*, uint8_t*, int);
  int block, gpl;
};

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 65536;
  const int N = argc > 2 ? atoi(argv[2]) : 2000;
  uint4* st;
  int* tok;
  uint8_t* done;
  CK(hipMalloc(&st, (size_t)B * 64));
  CK(hipMalloc(&tok, (size_t)B * 12));
  CK(hipMalloc(&done, B));
  std::vector<uint8_t> hs((size_t)B * 64), ht((size_t)B * 12);
  for (auto& x : hs) x = (uint8_t)((rand() % 5) - 2);
  for (size_t i = 0; i < ht.size(); ++i) ht[i] = (i / 12) % 2 ? 1 : (uint8_t)(rand() % 3);  // half the games get a no-op
  CK(hipMemcpy(tok, ht.data(), ht.size(), hipMemcpyHostToDevice));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  std::vector<Variant> vs = {
      {"empty  256thr x1", k_empty<256>, 256, 1},   {"empty 1024thr x1", k_empty<1024>, 1024, 1},
      {"copy   256thr x1", k_copy<256>, 256, 1},    {"copy  1024thr x1", k_copy<1024>, 1024, 1},
      {"step32 256thr x1", k_step32<256, 1>, 256, 1}, {"steppk 256thr x1", k_steppk<256>, 256, 1},
      {"copytok 256thr x1", k_copytok<256>, 256, 1},  {"copy nt-store 256", k_copy_nt<256, 0>, 256, 1},
      {"copy nt-ld+st 256", k_copy_nt<256, 1>, 256, 1},  {"steppk 512thr x1", k_steppk<512>, 512, 1},
      {"step32 256thr x2", k_step32<256, 2>, 256, 2},
      {"step32 256thr x4", k_step32<256, 4>, 256, 4}, {"step32 512thr x1", k_step32<512, 1>, 512, 1},
      {"step32 1024thr x1", k_step32<1024, 1>, 1024, 1}, {"step32 1024thr x2", k_step32<1024, 2>, 1024, 2},
      {"step32 1024thr x4", k_step32<1024, 4>, 1024, 4},
      {"stepb  256thr x1", k_stepb<256, 1>, 256, 1}, {"stepb  256thr x2", k_stepb<256, 2>, 256, 2},
      {"stepb  512thr x2", k_stepb<512, 2>, 512, 2}, {"stepb 1024thr x1", k_stepb<1024, 1>, 1024, 1},
      {"stepb 1024thr x2", k_stepb<1024, 2>, 1024, 2}, {"stepb 1024thr x4", k_stepb<1024, 4>, 1024, 4},
  };
  const bool quick = argc > 3 && atoi(argv[3]) != 0;
  if (quick) vs.resize(10);
  printf("B=%d games, %d launches per graph; algorithmic bytes per launch = %.2f MB\n", B, N, B * 141 / 1e6);
  for (int rep = 0; rep < 2; ++rep)
    for (auto& v : vs) {
      CK(hipMemcpy(st, hs.data(), hs.size(), hipMemcpyHostToDevice));
      const int threads = 4 * B / v.gpl;
      const int grid = (threads + v.block - 1) / v.block;
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(v.kern, dim3(grid), dim3(v.block), 0, s, st, st, tok, done, B);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      // eager launches for comparison
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(v.kern, dim3(grid), dim3(v.block), 0, s, st, st, tok, done, B);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms2;
      CK(hipEventElapsedTime(&ms2, e0, e1));
      if (rep == 1)
        printf("%-18s grid=%5d  graph %.3f us/launch (%.0f GB/s alg)   eager %.3f us/launch\n", v.name, grid,
               ms * 1e3 / N, B * 141.0 / (ms * 1e-3 / N) / 1e9, ms2 * 1e3 / N);
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
    }
  {
    FatArgs fa{st, st, tok, done, nullptr, nullptr, nullptr, B, 64, 64, 4, 1, 1};
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemcpy(st, hs.data(), hs.size(), hipMemcpyHostToDevice));
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_steppk_fat<256>, dim3(4 * B / 256), dim3(256), 0, s, fa);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("steppk 256thr, 96-byte kernarg struct: graph %.3f us/launch\n", ms * 1e3 / N);
    }
  }
  // the product entry point itself (libtensorgame.so), same harness; rep 2: a different token buffer every launch
  int* tok14;
  CK(hipMalloc(&tok14, (size_t)B * 12 * 14));
  for (int r = 0; r < 14; ++r) CK(hipMemcpy((char*)tok14 + (size_t)r * B * 12, ht.data(), ht.size(), hipMemcpyHostToDevice));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(st, hs.data(), hs.size(), hipMemcpyHostToDevice));
    uint8_t* ovf;
    CK(hipMalloc(&ovf, B));
    CK(hipMemset(ovf, 0, B));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i)
      if (tg_step_i8((const int8_t*)st, (int8_t*)st,
                     rep == 2 ? (const int8_t*)tok14 + (size_t)(i % 14) * B * 12 : (const int8_t*)tok, done,
                     rep ? ovf : nullptr, B, 4, 64, 1, s)) {
        fprintf(stderr, "tg_step_i8: %s\n", tg_last_error());
        return 1;
      }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("product tg_step_i8 (overflow %s)  graph %.3f us/launch (%.0f GB/s alg)\n", rep == 2 ? "tracked, 14 token buffers" : rep ? "tracked" : "NULL",
           ms * 1e3 / N, B * 141.0 / (ms * 1e-3 / N) / 1e9);
  }
  if (quick) return 0;
  // ---- split the batch into P independent chains inside ONE graph (fork/join by events) ----
  for (int P : {1, 2, 4, 8}) {
    CK(hipMemcpy(st, hs.data(), hs.size(), hipMemcpyHostToDevice));
    std::vector<hipStream_t> ss(P);
    std::vector<hipEvent_t> ev(P);
    for (int p = 0; p < P; ++p) {
      CK(hipStreamCreate(&ss[p]));
      CK(hipEventCreateWithFlags(&ev[p], hipEventDisableTiming));
    }
    hipEvent_t fork;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    const int Bp = B / P;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    CK(hipEventRecord(fork, s));
    for (int p = 0; p < P; ++p) CK(hipStreamWaitEvent(ss[p], fork, 0));
    for (int i = 0; i < N; ++i)
      for (int p = 0; p < P; ++p)
        if (tg_step_i8((const int8_t*)st + (size_t)p * Bp * 64, (int8_t*)st + (size_t)p * Bp * 64,
                       (const int8_t*)tok + (size_t)p * Bp * 12, done + (size_t)p * Bp, nullptr, Bp, 4, 64, 1, ss[p]))
          return 1;
    for (int p = 0; p < P; ++p) {
      CK(hipEventRecord(ev[p], ss[p]));
      CK(hipStreamWaitEvent(s, ev[p], 0));
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("product, %d independent chains of %d games: %.3f us per full-batch step (%.0f GB/s alg)\n", P, Bp,
           ms * 1e3 / N, B * 141.0 / (ms * 1e-3 / N) / 1e9);
  }
  // ---- P separate graphs on P separate streams (one chain of half/quarter-batch launches each), replayed
  //      concurrently: do the dependent-launch boundaries of different hardware queues overlap? ----
  for (int P : {1, 2, 4}) {
    CK(hipMemcpy(st, hs.data(), hs.size(), hipMemcpyHostToDevice));
    std::vector<hipStream_t> ss(P);
    std::vector<hipGraphExec_t> ges(P);
    std::vector<hipEvent_t> ev(P);
    const int Bp = B / P;
    for (int p = 0; p < P; ++p) {
      CK(hipStreamCreateWithFlags(&ss[p], hipStreamNonBlocking));
      CK(hipEventCreateWithFlags(&ev[p], hipEventDisableTiming));
      hipGraph_t g;
      CK(hipStreamBeginCapture(ss[p], hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < N; ++i)
        if (tg_step_i8((const int8_t*)st + (size_t)p * Bp * 64, (int8_t*)st + (size_t)p * Bp * 64,
                       (const int8_t*)tok + (size_t)p * Bp * 12, done + (size_t)p * Bp, nullptr, Bp, 4, 64, 1, ss[p]))
          return 1;
      CK(hipStreamEndCapture(ss[p], &g));
      CK(hipGraphInstantiate(&ges[p], g, nullptr, nullptr, 0));
    }
    hipEvent_t fork, e0, e1;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      CK(hipEventRecord(fork, s));
      for (int p = 0; p < P; ++p) {
        CK(hipStreamWaitEvent(ss[p], fork, 0));
        CK(hipGraphLaunch(ges[p], ss[p]));
        CK(hipEventRecord(ev[p], ss[p]));
      }
      for (int p = 0; p < P; ++p) CK(hipStreamWaitEvent(s, ev[p], 0));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("product, %d graphs on %d streams (%d games each): %.3f us per full-batch step (%.0f GB/s alg)\n", P, P, Bp,
           ms * 1e3 / N, B * 141.0 / (ms * 1e-3 / N) / 1e9);
  }
  return 0;
}
