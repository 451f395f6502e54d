// Standalone probe (no torch): what bounds ONE S=4 step launch at BASELINE config 4's per-GPU share
// (131 072 games = 8 MiB of states, 2 048 workgroups of 256 threads = 8 wavefronts per SIMD)?
// hipGraph replays of N chained in-place launches; per-launch time by HIP events.
//   empty<BLOCK>           launch/boundary floor for the same number of threads, by workgroup shape
//   copy<BLOCK,GPL,ST>     16-byte load + store per slice (+ done byte); GPL slices per lane; ST: 0 plain, 1 nt store
//   step<BLOCK,GPL,ST>     the packed int16 step (8 saturating v_pk_mad_i16 per slice), GPL slices per lane
//   product                tg_step_i8 of libtensorgame.so
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/s4_share_probe.hip -Lmat_mul_amd/lib -ltensorgame -o /tmp/s4_share_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int sbyte(uint32_t w, int t) { return __builtin_amdgcn_sbfe((int)w, 8 * t, 8); }
__device__ __forceinline__ uint32_t pack4(int n0, int n1, int n2, int n3) {
  uint32_t lo = __builtin_amdgcn_perm((uint32_t)n1, (uint32_t)n0, 0x0c0c0400u);
  uint32_t hi = __builtin_amdgcn_perm((uint32_t)n3, (uint32_t)n2, 0x0c0c0400u);
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}
__device__ __forceinline__ uint32_t pkmad_lo(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t pkmad_hi(uint32_t a, uint32_t b, uint32_t c) {
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

struct Args {
  const uint4* in;
  uint4* out;
  const int* tok;
  uint8_t* done;
  int B;
};

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_empty(Args a) {}

template <int ST>
__device__ __forceinline__ void store16(uint4* p, const uint4& q) {
  if constexpr (ST == 1) __builtin_nontemporal_store(v4u{q.x, q.y, q.z, q.w}, reinterpret_cast<v4u*>(p));
  else *p = q;
}

// workgroup w owns chunks [w*BLOCK*GPL, (w+1)*BLOCK*GPL); lane's slice n is chunk w*BLOCK*GPL + n*BLOCK + tid
template <int BLOCK, int GPL, int ST>
__global__ __launch_bounds__(BLOCK) void k_copy(Args a) {
  const int base = blockIdx.x * (BLOCK * GPL) + threadIdx.x;
  uint4 q[GPL];
#pragma unroll
  for (int n = 0; n < GPL; ++n) {
    int c = base + n * BLOCK;
    if (c >= 4 * a.B) c = 4 * a.B - 1;
    q[n] = a.in[c];
  }
#pragma unroll
  for (int n = 0; n < GPL; ++n) {
    const int c = base + n * BLOCK;
    if (c < 4 * a.B) {
      store16<ST>(a.out + c, q[n]);
      if ((c & 3) == 0) a.done[c >> 2] = (q[n].x | q[n].y | q[n].z | q[n].w) == 0;
    }
  }
}

// TOK: 0 = three dwords per lane from global memory (dwordx3), 1 = the workgroup's tokens staged through LDS
template <int BLOCK, int GPL, int ST, int TOK = 0>
__global__ __launch_bounds__(BLOCK) void k_step(Args a) {
  const int base = blockIdx.x * (BLOCK * GPL) + threadIdx.x;
  const int q = threadIdx.x & 3;
  uint4 pk[GPL];
  uint32_t tk[GPL][3];
  int cc[GPL];
  __shared__ uint32_t ltok[TOK ? BLOCK * GPL / 4 * 3 : 1];
  if constexpr (TOK == 1) {  // BLOCK*GPL/4 games x 12 bytes = BLOCK*GPL*3/4 dwords, contiguous
    const int g0 = blockIdx.x * (BLOCK * GPL / 4);
    for (int i = threadIdx.x; i < BLOCK * GPL / 4 * 3; i += BLOCK) {
      const long long idx = (long long)g0 * 3 + i;
      ltok[i] = idx < 3ll * a.B ? a.tok[idx] : 0;
    }
  }
#pragma unroll
  for (int n = 0; n < GPL; ++n) {
    int c = base + n * BLOCK;
    if (c >= 4 * a.B) c = 4 * a.B - 1;
    cc[n] = c;
    pk[n] = a.in[c];
    if constexpr (TOK == 0) {
      const int g = c >> 2;
      tk[n][0] = a.tok[g * 3];
      tk[n][1] = a.tok[g * 3 + 1];
      tk[n][2] = a.tok[g * 3 + 2];
    }
  }
  if constexpr (TOK == 1) {
    __syncthreads();
#pragma unroll
    for (int n = 0; n < GPL; ++n) {
      const int lg = (n * BLOCK + threadIdx.x) >> 2;
      tk[n][0] = ltok[lg * 3];
      tk[n][1] = ltok[lg * 3 + 1];
      tk[n][2] = ltok[lg * 3 + 2];
    }
  }
#pragma unroll
  for (int n = 0; n < GPL; ++n) {
    const uint32_t du = tk[n][0], dv = tk[n][1], dw = tk[n][2];
    const uint32_t shp = 0x00010001u;
    const int ui = 1 - __builtin_amdgcn_sbfe((int)du, 8 * q, 8);
    const uint32_t uip = __builtin_amdgcn_perm((uint32_t)ui, (uint32_t)ui, 0x05040100u);
    const uint32_t yv = dv << 8, yw = dw << 8;
    const uint32_t vA = pksub(__builtin_amdgcn_perm(dv, yv, 0x0A050804u), shp);
    const uint32_t vB = pksub(__builtin_amdgcn_perm(dv, yv, 0x0B070906u), shp);
    const uint32_t wA = pksub(__builtin_amdgcn_perm(dw, yw, 0x0A050804u), shp);
    const uint32_t wB = pksub(__builtin_amdgcn_perm(dw, yw, 0x0B070906u), shp);
    const uint32_t uvA = pkmul_sat(vA, uip), uvB = pkmul_sat(vB, uip);
    const uint32_t x[4] = {pk[n].x, pk[n].y, pk[n].z, pk[n].w};
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
      a.done[0] = 2;  // stand-in for the overflow store
    }
    const uint64_t m = __ballot(nz != 0);
    const int lane = threadIdx.x & 63;
    const bool any = ((m >> (lane & ~3)) & 0xf) != 0;
    const int c = base + n * BLOCK;
    if (c < 4 * a.B) {
      store16<ST>(a.out + cc[n], uint4{o[0], o[1], o[2], o[3]});
      if (q == 0) a.done[c >> 2] = !any;
    }
  }
}


// copy + the token loads (no arithmetic): what the memory side of a step costs.  TOK: 0 = dwordx3 per lane,
// 2 = ONE dword per lane (lane q of a game loads token dword q; q = 3 re-reads dword 2) + quad broadcasts
template <int BLOCK, int TOK>
__global__ __launch_bounds__(BLOCK) void k_copytok(Args a) {
  const int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t < 4 * a.B) {
    uint4 q = a.in[t];
    const int g = t >> 2;
    uint32_t x, y, z;
    if constexpr (TOK == 0) {
      x = a.tok[g * 3], y = a.tok[g * 3 + 1], z = a.tok[g * 3 + 2];
    } else {
      const int qq = t & 3;
      const uint32_t mine = a.tok[g * 3 + (qq < 3 ? qq : 2)];
      x = __builtin_amdgcn_mov_dpp(mine, 0x00, 0xf, 0xf, true);
      y = __builtin_amdgcn_mov_dpp(mine, 0x55, 0xf, 0xf, true);
      z = __builtin_amdgcn_mov_dpp(mine, 0xAA, 0xf, 0xf, true);
    }
    a.out[t] = q;
    if ((t & 3) == 0) a.done[g] = ((q.x | q.y | q.z | q.w) == 0) | ((x ^ y ^ z) == 0x7fffffff);
  }
}

// the packed step again with (TOK = 2) one token dword per lane + quad broadcasts and (BIAS) the state unpacked as
// b + 128 in [0, 255]: the int8 range test is "high byte of every int16 result is zero" (ORs only)
template <int BLOCK, int TOK, bool BIAS, int ST = 0>
__global__ __launch_bounds__(BLOCK) void k_step2(Args a) {
  const int t = blockIdx.x * BLOCK + threadIdx.x;
  const int q = threadIdx.x & 3;
  int c = t;
  if (c >= 4 * a.B) c = 4 * a.B - 1;
  const uint4 pk = a.in[c];
  const int g = c >> 2;
  uint32_t du, dv, dw;
  if constexpr (TOK == 0) {
    du = a.tok[g * 3], dv = a.tok[g * 3 + 1], dw = a.tok[g * 3 + 2];
  } else {
    const uint32_t mine = a.tok[g * 3 + (q < 3 ? q : 2)];
    du = __builtin_amdgcn_mov_dpp(mine, 0x00, 0xf, 0xf, true);
    dv = __builtin_amdgcn_mov_dpp(mine, 0x55, 0xf, 0xf, true);
    dw = __builtin_amdgcn_mov_dpp(mine, 0xAA, 0xf, 0xf, true);
  }
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
    uint32_t lo, hi;
    if constexpr (BIAS) {
      const uint32_t xb = x[d] ^ 0x80808080u;
      lo = __builtin_amdgcn_perm(0u, xb, 0x0c010c00u);
      hi = __builtin_amdgcn_perm(0u, xb, 0x0c030c02u);
    } else {
      const uint32_t y = x[d] << 8;
      lo = __builtin_amdgcn_perm(x[d], y, 0x0A050804u), hi = __builtin_amdgcn_perm(x[d], y, 0x0B070906u);
    }
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
    if constexpr (BIAS) {
      ovf |= A[2 * d] | A[2 * d + 1];
      o[d] = __builtin_amdgcn_perm(A[2 * d + 1], A[2 * d], 0x06040200u) ^ 0x80808080u;
    } else {
      ovf |= pkaddu(A[2 * d], 0x00800080u) | pkaddu(A[2 * d + 1], 0x00800080u);
      o[d] = __builtin_amdgcn_perm(A[2 * d + 1], A[2 * d], 0x06040200u);
    }
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
    a.done[0] = 2;  // stand-in for the overflow store
  }
  const uint64_t m = __ballot(nz != 0);
  const int lane = threadIdx.x & 63;
  const bool any = ((m >> (lane & ~3)) & 0xf) != 0;
  if (t < 4 * a.B) {
    store16<ST>(a.out + c, uint4{o[0], o[1], o[2], o[3]});
    if (q == 0) a.done[g] = !any;
  }
}

// Third generation: BLOCK threads, TOK: 0 = dwordx3 per lane, 1 = the workgroup's tokens staged through LDS by
// coalesced dword loads, 2 = one dword per lane + quad broadcasts; biased unpack always; NTL: non-temporal state loads;
// SF: the state load is issued FIRST and unpacked while the tokens are still on their way.
template <int BLOCK, int TOK, bool NTL, bool SF>
__global__ __launch_bounds__(BLOCK) void k_step3(Args a) {
  const int t = blockIdx.x * BLOCK + threadIdx.x;
  const int q = threadIdx.x & 3;
  int c = t;
  if (c >= 4 * a.B) c = 4 * a.B - 1;
  const int g = c >> 2;
  __shared__ uint32_t ltok[TOK == 1 ? BLOCK / 4 * 3 : 1];
  uint4 pk;
  auto load_state = [&]() {
    if constexpr (NTL) {
      const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(a.in + c));
      pk = uint4{v.x, v.y, v.z, v.w};
    } else {
      pk = a.in[c];
    }
  };
  uint32_t du = 0, dv = 0, dw = 0, mine = 0;
  auto load_tok = [&]() {
    if constexpr (TOK == 0) {
      du = a.tok[g * 3], dv = a.tok[g * 3 + 1], dw = a.tok[g * 3 + 2];
    } else if constexpr (TOK == 2) {
      mine = a.tok[g * 3 + (q < 3 ? q : 2)];
    } else {
      const int g0 = blockIdx.x * (BLOCK / 4);
      if (threadIdx.x < BLOCK / 4 * 3) {
        const long long idx = (long long)g0 * 3 + threadIdx.x;
        mine = idx < 3ll * a.B ? a.tok[idx] : 0;
      }
    }
  };
  if constexpr (SF) { load_state(); load_tok(); } else { load_tok(); load_state(); }
  uint32_t lo[4], hi[4];
  auto unpack = [&]() {
    const uint32_t x[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t xb = x[d] ^ 0x80808080u;
      lo[d] = __builtin_amdgcn_perm(0u, xb, 0x0c010c00u);
      hi[d] = __builtin_amdgcn_perm(0u, xb, 0x0c030c02u);
    }
  };
  if constexpr (SF) {
    unpack();
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (TOK == 2) {
    du = __builtin_amdgcn_mov_dpp(mine, 0x00, 0xf, 0xf, true);
    dv = __builtin_amdgcn_mov_dpp(mine, 0x55, 0xf, 0xf, true);
    dw = __builtin_amdgcn_mov_dpp(mine, 0xAA, 0xf, 0xf, true);
  } else if constexpr (TOK == 1) {
    if (threadIdx.x < BLOCK / 4 * 3) ltok[threadIdx.x] = mine;
    __syncthreads();
    const int lg = threadIdx.x >> 2;
    du = ltok[lg * 3], dv = ltok[lg * 3 + 1], dw = ltok[lg * 3 + 2];
  }
  const uint32_t shp = 0x00010001u;
  const int ui = 1 - __builtin_amdgcn_sbfe((int)du, 8 * q, 8);
  const uint32_t uip = __builtin_amdgcn_perm((uint32_t)ui, (uint32_t)ui, 0x05040100u);
  const uint32_t yv = dv << 8, yw = dw << 8;
  const uint32_t vA = pksub(__builtin_amdgcn_perm(dv, yv, 0x0A050804u), shp);
  const uint32_t vB = pksub(__builtin_amdgcn_perm(dv, yv, 0x0B070906u), shp);
  const uint32_t wA = pksub(__builtin_amdgcn_perm(dw, yw, 0x0A050804u), shp);
  const uint32_t wB = pksub(__builtin_amdgcn_perm(dw, yw, 0x0B070906u), shp);
  const uint32_t uvA = pkmul_sat(vA, uip), uvB = pkmul_sat(vB, uip);
  if constexpr (!SF) unpack();
  uint32_t A[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t uv = d < 2 ? uvA : uvB;
    if (d & 1) {
      A[2 * d] = pkmad_hi(uv, wA, lo[d]);
      A[2 * d + 1] = pkmad_hi(uv, wB, hi[d]);
    } else {
      A[2 * d] = pkmad_lo(uv, wA, lo[d]);
      A[2 * d + 1] = pkmad_lo(uv, wB, hi[d]);
    }
  }
  uint32_t o[4], nz = 0, ovf = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    ovf |= A[2 * d] | A[2 * d + 1];
    o[d] = __builtin_amdgcn_perm(A[2 * d + 1], A[2 * d], 0x06040200u) ^ 0x80808080u;
    nz |= o[d];
  }
  if (ovf & 0xFF00FF00u) {  // rare: exact 32-bit redo of this lane's slice
    const uint32_t x[4] = {pk.x, pk.y, pk.z, pk.w};
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
    a.done[0] = 2;  // stand-in for the overflow store
  }
  const uint64_t m = __ballot(nz != 0);
  const int lane = threadIdx.x & 63;
  const bool any = ((m >> (lane & ~3)) & 0xf) != 0;
  if (t < 4 * a.B) {
    a.out[c] = uint4{o[0], o[1], o[2], o[3]};
    if (q == 0) a.done[g] = !any;
  }
}

// Fourth generation (dpp tokens, biased unpack): store policy and load ordering.
//   ST: 0 plain global store, 1 nt, 2 buffer store sc1, 3 buffer store sc0 sc1, 4 buffer store sc0
//   TW: the lane WAITS for its tokens before it requests the state (throttles the state requests in flight)
//   NTL: non-temporal state loads
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned bu4;
template <int BLOCK, int ST, bool TW, bool NTL, bool COPYONLY = false>
__global__ __launch_bounds__(BLOCK) void k_step4(Args a) {
  const int t = blockIdx.x * BLOCK + threadIdx.x;
  const int q = threadIdx.x & 3;
  int c = t;
  if (c >= 4 * a.B) c = 4 * a.B - 1;
  const int g = c >> 2;
  uint32_t mine = 0;
  if constexpr (!COPYONLY) mine = a.tok[g * 3 + (q < 3 ? q : 2)];
  if constexpr (TW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint4 pk;
  if constexpr (NTL) {
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(a.in + c));
    pk = uint4{v.x, v.y, v.z, v.w};
  } else {
    pk = a.in[c];
  }
  uint32_t o[4] = {pk.x, pk.y, pk.z, pk.w}, nz = pk.x | pk.y | pk.z | pk.w;
  if constexpr (!COPYONLY) {
    const uint32_t du = __builtin_amdgcn_mov_dpp(mine, 0x00, 0xf, 0xf, true);
    const uint32_t dv = __builtin_amdgcn_mov_dpp(mine, 0x55, 0xf, 0xf, true);
    const uint32_t dw = __builtin_amdgcn_mov_dpp(mine, 0xAA, 0xf, 0xf, true);
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
      const uint32_t xb = x[d] ^ 0x80808080u;
      const uint32_t lo = __builtin_amdgcn_perm(0u, xb, 0x0c010c00u), hi = __builtin_amdgcn_perm(0u, xb, 0x0c030c02u);
      const uint32_t uv = d < 2 ? uvA : uvB;
      if (d & 1) {
        A[2 * d] = pkmad_hi(uv, wA, lo);
        A[2 * d + 1] = pkmad_hi(uv, wB, hi);
      } else {
        A[2 * d] = pkmad_lo(uv, wA, lo);
        A[2 * d + 1] = pkmad_lo(uv, wB, hi);
      }
    }
    uint32_t ovf = 0;
    nz = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      ovf |= A[2 * d] | A[2 * d + 1];
      o[d] = __builtin_amdgcn_perm(A[2 * d + 1], A[2 * d], 0x06040200u) ^ 0x80808080u;
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
      a.done[0] = 2;
    }
  }
  const uint64_t m = __ballot(nz != 0);
  const int lane = threadIdx.x & 63;
  const bool any = ((m >> (lane & ~3)) & 0xf) != 0;
  if (t < 4 * a.B) {
    if constexpr (ST <= 1) {
      store16<ST>(a.out + c, uint4{o[0], o[1], o[2], o[3]});
    } else {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7fffffff, 0x00027000);
      __builtin_amdgcn_raw_buffer_store_b128(bu4{o[0], o[1], o[2], o[3]}, rs, c * 16, 0, ST == 2 ? 16 : (ST == 3 ? 17 : 1));
    }
    if (q == 0) a.done[g] = !any;
  }
}

struct Variant {
  const char* name;
  void (*kern)(Args);
  int block, gpl;
};

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 131072;
  const int N = argc > 2 ? atoi(argv[2]) : 2000;
  const int NTOK = argc > 3 ? atoi(argv[3]) : 1;  // token buffers cycled through by every variant (bench.py uses 14)
  const char* only = argc > 4 ? argv[4] : nullptr;  // run only the variants whose name contains this
  uint4* st;
  int* tok;
  uint8_t* done;
  CK(hipMalloc(&st, (size_t)B * 64));
  CK(hipMalloc(&tok, (size_t)B * 12 * NTOK));
  CK(hipMalloc(&done, B));
  std::vector<uint8_t> hs((size_t)B * 64), ht((size_t)B * 12);
  for (auto& x : hs) x = (uint8_t)((rand() % 5) - 2);
  for (size_t i = 0; i < ht.size(); ++i) ht[i] = (i / 12) % 2 ? 1 : (uint8_t)(rand() % 3);  // half the games get a no-op
  for (int r = 0; r < NTOK; ++r) CK(hipMemcpy((char*)tok + (size_t)r * B * 12, ht.data(), ht.size(), hipMemcpyHostToDevice));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  std::vector<Variant> vs = {
      {"empty  256thr", k_empty<256>, 256, 1},
      {"empty  512thr", k_empty<512>, 512, 1},
      {"empty 1024thr", k_empty<1024>, 1024, 1},
      {"empty  256thr half the threads", k_empty<256>, 256, 2},
      {"empty  256thr quarter", k_empty<256>, 256, 4},
      {"copy  256thr x1", k_copy<256, 1, 0>, 256, 1},
      {"copy  256thr x2", k_copy<256, 2, 0>, 256, 2},
      {"copy  256thr x4", k_copy<256, 4, 0>, 256, 4},
      {"copy  512thr x1", k_copy<512, 1, 0>, 512, 1},
      {"copy 1024thr x1", k_copy<1024, 1, 0>, 1024, 1},
      {"copy  512thr x2", k_copy<512, 2, 0>, 512, 2},
      {"copy  256thr x1 nt-store", k_copy<256, 1, 1>, 256, 1},
      {"copy  256thr x2 nt-store", k_copy<256, 2, 1>, 256, 2},
      {"step  256thr x1", k_step<256, 1, 0>, 256, 1},
      {"step  256thr x2", k_step<256, 2, 0>, 256, 2},
      {"step  256thr x4", k_step<256, 4, 0>, 256, 4},
      {"step  512thr x1", k_step<512, 1, 0>, 512, 1},
      {"step  512thr x2", k_step<512, 2, 0>, 512, 2},
      {"step 1024thr x1", k_step<1024, 1, 0>, 1024, 1},
      {"step 1024thr x2", k_step<1024, 2, 0>, 1024, 2},
      {"step  256thr x1 nt-store", k_step<256, 1, 1>, 256, 1},
      {"step  256thr x2 nt-store", k_step<256, 2, 1>, 256, 2},
      {"copytok 256thr dwordx3", k_copytok<256, 0>, 256, 1},
      {"copytok 256thr dword+dpp", k_copytok<256, 2>, 256, 1},
      {"step2 256thr (= step x1)", k_step2<256, 0, false>, 256, 1},
      {"step2 256thr dpp-tok", k_step2<256, 2, false>, 256, 1},
      {"step2 256thr biased", k_step2<256, 0, true>, 256, 1},
      {"step2 256thr dpp-tok biased", k_step2<256, 2, true>, 256, 1},
      {"step2 512thr dpp-tok biased", k_step2<512, 2, true>, 512, 1},
      {"step3 256 dx3", k_step3<256, 0, false, false>, 256, 1},
      {"step3 256 dx3 state-first", k_step3<256, 0, false, true>, 256, 1},
      {"step3 256 dpp", k_step3<256, 2, false, false>, 256, 1},
      {"step3 256 dpp state-first", k_step3<256, 2, false, true>, 256, 1},
      {"step3 256 lds", k_step3<256, 1, false, false>, 256, 1},
      {"step3 256 lds state-first", k_step3<256, 1, false, true>, 256, 1},
      {"step3 512 lds", k_step3<512, 1, false, false>, 512, 1},
      {"step3 512 dpp", k_step3<512, 2, false, false>, 512, 1},
      {"step3 1024 lds", k_step3<1024, 1, false, false>, 1024, 1},
      {"step3 1024 dpp", k_step3<1024, 2, false, false>, 1024, 1},
      {"step3 256 dx3 nt", k_step3<256, 0, true, false>, 256, 1},
      {"step3 256 lds nt", k_step3<256, 1, true, false>, 256, 1},
      {"step3 256 dpp nt", k_step3<256, 2, true, false>, 256, 1},
      {"step3 512 lds nt", k_step3<512, 1, true, false>, 512, 1},
      {"step3 512 dpp nt", k_step3<512, 2, true, false>, 512, 1},
      {"step3 1024 lds nt", k_step3<1024, 1, true, false>, 1024, 1},
      {"step3 1024 dpp nt", k_step3<1024, 2, true, false>, 1024, 1},
      {"copy4 plain", k_step4<256, 0, false, false, true>, 256, 1},
      {"copy4 store sc1", k_step4<256, 2, false, false, true>, 256, 1},
      {"copy4 store sc0 sc1", k_step4<256, 3, false, false, true>, 256, 1},
      {"copy4 store sc0", k_step4<256, 4, false, false, true>, 256, 1},
      {"step4 plain", k_step4<256, 0, false, false>, 256, 1},
      {"step4 store nt", k_step4<256, 1, false, false>, 256, 1},
      {"step4 store sc1", k_step4<256, 2, false, false>, 256, 1},
      {"step4 store sc0 sc1", k_step4<256, 3, false, false>, 256, 1},
      {"step4 store sc0", k_step4<256, 4, false, false>, 256, 1},
      {"step4 token-wait", k_step4<256, 0, true, false>, 256, 1},
      {"step4 token-wait nt-load", k_step4<256, 0, true, true>, 256, 1},
      {"step4 nt-load", k_step4<256, 0, false, true>, 256, 1},
      {"step4 512 token-wait", k_step4<512, 0, true, false>, 512, 1},
      {"step4 512 token-wait nt-load", k_step4<512, 0, true, true>, 512, 1},
      {"step  256thr x1 lds-tok", k_step<256, 1, 0, 1>, 256, 1},
      {"step  256thr x2 lds-tok", k_step<256, 2, 0, 1>, 256, 2},
      {"step  512thr x2 lds-tok", k_step<512, 2, 0, 1>, 512, 2},
  };
  printf("B=%d games, %d launches per graph, %d token buffers; algorithmic bytes per launch = %.2f MB\n", B, N, NTOK, B * 141 / 1e6);
  for (int rep = 0; rep < 2; ++rep)
    for (auto& v : vs) {
      if (only && !strstr(v.name, only)) continue;
      if (strstr(v.name, "store sc") && (size_t)B * 64 > 0x7fffffffu) continue;  // 32-bit buffer offsets
      CK(hipMemcpy(st, hs.data(), hs.size(), hipMemcpyHostToDevice));
      const int threads = 4 * B / v.gpl;
      const int grid = (threads + v.block - 1) / v.block;
      Args a{st, st, tok, done, B};
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
      for (int i = 0; i < N; ++i) {
        a.tok = tok + (size_t)(i % NTOK) * B * 3;
        hipLaunchKernelGGL(v.kern, dim3(grid), dim3(v.block), 0, s, a);
      }
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      float best = 1e30f;
      for (int t = 0; t < 3; ++t) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      if (rep == 1)
        printf("%-32s grid=%5d  %.3f us/launch (%.0f GB/s alg)\n", v.name, grid, best * 1e3 / N,
               B * 141.0 / (best * 1e-3 / N) / 1e9);
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
    }
  int* tok14;
  uint8_t* ovfb;
  CK(hipMalloc(&tok14, (size_t)B * 12 * 14));
  CK(hipMalloc(&ovfb, B));
  CK(hipMemset(ovfb, 0, B));
  for (int r = 0; r < 14; ++r) CK(hipMemcpy((char*)tok14 + (size_t)r * B * 12, ht.data(), ht.size(), hipMemcpyHostToDevice));
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemcpy(st, hs.data(), hs.size(), hipMemcpyHostToDevice));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i)
      if (tg_step_i8((const int8_t*)st, (int8_t*)st, rep >= 2 ? (const int8_t*)tok14 + (size_t)(i % 14) * B * 12 : (const int8_t*)tok, done,
                     rep >= 2 ? ovfb : nullptr, B, 4, 64, 1, s)) {
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
    float best = 1e30f;
    for (int t = 0; t < 3; ++t) {
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    if (rep & 1) printf("%-32s              %.3f us/launch (%.0f GB/s alg)\n", rep == 3 ? "product, 14 token buffers + ovf" : "product tg_step_i8", best * 1e3 / N, B * 141.0 / (best * 1e-3 / N) / 1e9);
  }
  return 0;
}
