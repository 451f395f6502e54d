// Generator side of libtensorgame.so: counter-based RNG -> factor tokens, random unimodular
// bases, and the change of basis (three mode products).  gfx950 only.
//
// The generator is split in two kernels: gen_tokens_kernel draws the factor vectors (Philox,
// per-vector rejection of the zero vector, optional change of basis on the factors) and writes
// the int8 tokens; the rank-1 accumulation then runs in the GENF kernels of tg_kernels.hip on
// those tokens (tg_gen_from_factors_i8).  The target is therefore always exactly the sum of the
// rank-1 terms of the EMITTED tokens.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "../../include/tensor_game.h"
#include "tg_device.h"

int tg_internal_fail(int code, const char* fmt, ...);  // tg_kernels.hip
namespace tg { struct Dist; }
// the generator in one kernel (tg_genfused.h, launched from tg_kernels.hip): 1 = launched, 0 = not applicable
int tg_internal_gen_fused(int8_t* target, int8_t* actions, uint8_t* overflow, const int8_t* basis, int64_t B, int S,
                          int R, const tg::Dist& D, int shift, uint64_t seed, uint64_t gid0, int64_t stride,
                          hipStream_t st);

namespace tg {

// one 32-bit draw -> value (the basis sampler: one draw per matrix cell)
__device__ __forceinline__ int draw_value(uint32_t d, const Dist& D) {
  if (D.nv == 3) {  // the reference's vocabulary (-1,0,1): two compares, two selects (wave-uniform branch)
    const int v = d >= D.thr[0] ? D.val[1] : D.val[0];
    return d >= D.thr[1] ? D.val[2] : v;
  }
  int idx = 0;
#pragma unroll
  for (int t = 0; t < TG_MAX_VALUES - 1; ++t) idx += (t < D.nv - 1) && (d >= D.thr[t]);
  int v = D.val[0];
#pragma unroll
  for (int t = 1; t < TG_MAX_VALUES; ++t) v = (idx == t) ? D.val[t] : v;
  return v;
}

// 16-byte chunk of a game from / to global memory; the game's last chunk holds only TAIL bytes
template <int TAIL>
__device__ __forceinline__ uint4 load_chunk16(const int8_t* p, bool tail) {
  if (TAIL != 0 && tail) {
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < TAIL; ++t) w[t >> 2] |= static_cast<uint32_t>(static_cast<uint8_t>(p[t])) << (8 * (t & 3));
    return uint4{w[0], w[1], w[2], w[3]};
  }
  return *reinterpret_cast<const uint4*>(p);
}
template <int TAIL>
__device__ __forceinline__ void store_chunk16(int8_t* p, const uint4& q, bool tail) {
  if (TAIL != 0 && tail) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int t = 0; t < TAIL; ++t) p[t] = static_cast<int8_t>(w[t >> 2] >> (8 * (t & 3)));
    return;
  }
  *reinterpret_cast<uint4*>(p) = q;
}

constexpr uint32_t kStreamBasis = 0x80000000u;
constexpr uint32_t kMaxAttempts = 1u << 16;

// Factor vectors (game b, term r, x in {u,v,w}).  Counter = (gid_lo, gid_hi, 3r+x,
// attempt<<8 | block), key = seed; a block yields eight 16-bit draws (elements 8 block + 2 word + half):
// identical to oracle/tensor_game.py::_draw_vector.
// ST > 0: S is a compile-time constant, the vector lives in registers.
// Rejection sampling diverges: at S=4 a vector is rejected with probability 0.24, and a wavefront
// that gives each lane ONE vector runs max-over-64-lanes attempts (about 3.9 instead of 1.3).  So a
// lane owns M vectors (idx = base + lane + 256*m) and walks them in ONE flat loop -- every trip is
// one attempt for whatever vector the lane is on -- and the rejections average out over M.
// The M*256 vectors of a workgroup are contiguous in actions_out: they are assembled in LDS and
// written with 16-byte stores.
template <int ST, int M>
__global__ __launch_bounds__(kBlock) void gen_tokens_kernel(int8_t* actions_out, uint8_t* overflow, int64_t B,
                                                            int Srt, int R, Dist D, int shift, uint64_t seed,
                                                            uint64_t gid0, const int8_t* basis, int vec16) {
  constexpr int SMAX = ST ? ST : TG_MAX_S;
  constexpr int NBLK = (SMAX + 7) / 8;  // one Philox block = eight 16-bit draws
  __shared__ __attribute__((aligned(16))) int8_t stage[kBlock * M * SMAX];
  const int S = ST ? ST : Srt;
  const int64_t nvec = B * R * 3;
  const uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
  const int64_t per_wg = static_cast<int64_t>(kBlock) * M;
  const int64_t nround = (nvec + gridDim.x * per_wg - 1) / (gridDim.x * per_wg);
  const uint32_t R3 = static_cast<uint32_t>(3 * R);
  for (int64_t it = 0; it < nround; ++it) {
    const int64_t base = (it * gridDim.x + blockIdx.x) * per_wg;  // first vector of this workgroup
    const int64_t b0 = base / R3;                                  // 64-bit division on the scalar unit
    const uint32_t within0 = static_cast<uint32_t>(base - b0 * R3);
    int m = 0;
    uint32_t attempt = 0;
    while (m < M) {
      const int64_t idx = base + threadIdx.x + static_cast<int64_t>(kBlock) * m;
      if (idx >= nvec) break;
      const uint32_t within = within0 + threadIdx.x + kBlock * m;
      const uint32_t db = within / R3;
      const int64_t b = b0 + db;
      const int sub = static_cast<int>(within - db * R3);  // 3r + x
      const uint64_t gid = gid0 + static_cast<uint64_t>(b);
      int f[SMAX];
      bool any = false;
#pragma unroll
      for (int q = 0; q < NBLK; ++q) {
        if (8 * q < S) {
          const U4 o = philox4x32_10(U4{static_cast<uint32_t>(gid), static_cast<uint32_t>(gid >> 32),
                                        static_cast<uint32_t>(sub), (attempt << 8) | static_cast<uint32_t>(q)},
                                     k0, k1);
          const uint32_t d[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int t = 0; t < 8; ++t) {  // element 8q + t: word t/2, low half first
            const int e = 8 * q + t;
            if (e < SMAX) {
              f[e] = (e < S) ? draw_value16((d[t >> 1] >> (16 * (t & 1))) & 0xFFFFu, D) : 0;
              any |= f[e] != 0;
            }
          }
        }
      }
      if (!any && attempt + 1 < kMaxAttempts) {
        ++attempt;
        continue;
      }
      int bad = 0;
      int8_t* dst = stage + (threadIdx.x + kBlock * m) * S;
      if (basis) {
        const int x = sub % 3;
        const int8_t* Mx = basis + (b * 3 + x) * S * S;
        for (int a = 0; a < S; ++a) {
          int acc = 0;
#pragma unroll
          for (int i = 0; i < SMAX; ++i)
            if (i < S) acc += Mx[a * S + i] * f[i];
          const int tokv = acc + shift;
          bad |= tokv + 128;
          dst[a] = static_cast<int8_t>(tokv);
        }
      } else {
#pragma unroll
        for (int e = 0; e < SMAX; ++e)
          if (e < S) {
            const int tokv = f[e] + shift;
            bad |= tokv + 128;
            dst[e] = static_cast<int8_t>(tokv);
          }
      }
      if (overflow && (bad & ~255)) overflow[b] = 1;
      ++m;
      attempt = 0;
    }
    __syncthreads();
    // the workgroup's vectors are the contiguous bytes [base*S, base*S + nlive*S) of actions_out
    const int64_t nlive = min(per_wg, nvec - base);
    const int nbytes = nlive > 0 ? static_cast<int>(nlive) * S : 0;
    int8_t* out = actions_out + base * S;
    int body = 0;
    if (vec16) {
      body = nbytes & ~15;
      for (int o = 16 * threadIdx.x; o < body; o += 16 * kBlock)
        *reinterpret_cast<uint4*>(out + o) = *reinterpret_cast<const uint4*>(stage + o);
    }
    for (int o = body + threadIdx.x; o < nbytes; o += kBlock) out[o] = stage[o];
    __syncthreads();
  }
}

// Change of basis on the FACTORS, in place on the token array: for game b and mode x the R vectors
// f_r (tokens - shift) become M_x f_r.  One workgroup of three wavefronts per game -- wavefront x
// does mode x.  The game's token block (R*3S contiguous bytes, tiled by RT actions) is staged
// through LDS so that global reads and writes are coalesced; lane r keeps its vector f_r in
// registers, and the matrix entries M_x[a][i] are the same for every lane: LDS broadcast reads.
// Tokens that leave int8 wrap and raise the game's overflow flag (the target is then built from
// the emitted, wrapped tokens).  ST = 0: runtime S (vector in scratch; correctness fallback).
template <int ST>
__global__ __launch_bounds__(192) void basis_tokens_kernel(int8_t* actions, uint8_t* overflow, int64_t B, int Srt,
                                                           int R, int shift, const int8_t* basis) {
  constexpr int SMAX = ST ? ST : TG_MAX_S;
  constexpr int RT = 64;  // actions per tile: one per lane
  extern __shared__ __attribute__((aligned(16))) int bt_lds[];
  const int S = ST ? ST : Srt;
  int* Mall = bt_lds;                                           // 3*S*S ints
  int8_t* T = reinterpret_cast<int8_t*>(bt_lds + 3 * S * S);    // RT*3S token bytes
  const int x = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const int8_t* Msrc = basis + b * 3 * S * S;
    for (int c = threadIdx.x; c < 3 * S * S; c += 192) Mall[c] = Msrc[c];
    int bad = 0;
    for (int r0 = 0; r0 < R; r0 += RT) {
      const int nr = min(RT, R - r0);
      int8_t* blk = actions + (b * R + r0) * 3 * S;  // nr*3S contiguous bytes
      const int nbytes = nr * 3 * S;
      __syncthreads();
      for (int o = threadIdx.x; o < nbytes; o += 192) T[o] = blk[o];
      __syncthreads();
      if (lane < nr) {
        int8_t* v = T + lane * 3 * S + x * S;
        const int* M = Mall + x * S * S;
        int f[SMAX];
#pragma unroll
        for (int i = 0; i < SMAX; ++i) f[i] = (i < S) ? v[i] - shift : 0;
        for (int a = 0; a < S; ++a) {
          int acc = 0;
#pragma unroll
          for (int i = 0; i < SMAX; ++i)
            if (i < S) acc += M[a * S + i] * f[i];
          const int tokv = acc + shift;
          bad |= tokv + 128;
          v[a] = static_cast<int8_t>(tokv);
        }
      }
      __syncthreads();
      for (int o = threadIdx.x; o < nbytes; o += 192) blk[o] = T[o];
    }
    bad = __syncthreads_or(bad & ~255);
    if (threadIdx.x == 0 && overflow && bad) overflow[b] = 1;
  }
}

// The same change of basis on the matrix cores (S = 9, 16, 25).  For one game, mode x and 32 actions
//     D[a][r] = sum_i M_x[a][i] * t_r[i]          (t = raw token bytes, i < S; 32x32x32 int8 MFMA)
//     token'_r[a] = D[a][r] - shift * rowsum(M_x[a]) + shift
// The A fragment is a row of M_x, the B fragment a row of the token block: both are 16 contiguous
// bytes per lane, read straight from global memory (unaligned dwords, nothing past the S valid bytes
// of a row is touched; bytes i >= S of both fragments are zero, so padding never contributes).  The
// result has the action on the lane and four consecutive a per register group: one dword store.
// A job reads only the bytes it later writes, so the transform is in place.  Exact: int32 sums; tokens
// that leave int8 wrap and raise the game's overflow flag, as in basis_tokens_kernel.
typedef int bt_v4i __attribute__((ext_vector_type(4)));
typedef int bt_v16i __attribute__((ext_vector_type(16)));

template <int S>
__device__ __forceinline__ bt_v4i load_row_fragment(const int8_t* row, int h) {  // bytes 16h..16h+15 of an S-byte row
  bt_v4i f;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t w = 0;
    const int k0 = 16 * h + 4 * d;
    if (k0 + 3 < S) {
      __builtin_memcpy(&w, row + k0, 4);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (k0 + t < S) w |= static_cast<uint32_t>(static_cast<uint8_t>(row[k0 + t])) << (8 * t);
    }
    f[d] = static_cast<int>(w);
  }
  return f;
}

template <int S>
__global__ __launch_bounds__(512) void basis_tokens_mfma_kernel(int8_t* actions, uint8_t* overflow, int64_t B, int R,
                                                                int shift, const int8_t* basis) {
  // launched with one wavefront per job of a game when there are at most 8 (3 modes x ceil(R/32) action tiles:
  // 6 at R = 64), so that no wavefront does two jobs while others idle
  __shared__ __attribute__((aligned(16))) int rowsum[8][32];
  const int nwave = static_cast<int>(blockDim.x >> 6);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int ntile = (R + 31) >> 5, njob = 3 * ntile;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    int8_t* blk = actions + b * R * 3 * S;
    int hi = 0, lo = 0;
    for (int job = wave; job < njob; job += nwave) {
      const int x = job / ntile, r0 = 32 * (job - x * ntile);
      // A: row a = col of M_x (rows >= S shadow the last one; their results are never used)
      const int8_t* mrow = basis + ((b * 3 + x) * S + (col < S ? col : S - 1)) * S;
      const bt_v4i fa = load_row_fragment<S>(mrow, h);
      // B: action r0 + col (actions >= R shadow the last one; never stored)
      const int r = r0 + col;
      int8_t* trow = blk + static_cast<int64_t>(r < R ? r : R - 1) * 3 * S + x * S;
      const bt_v4i fb = load_row_fragment<S>(trow, h);
      // row sums of M_x for the shift correction: this lane has half a row, its partner lane the other half
      int part = 0;
#pragma unroll
      for (int d = 0; d < 4; ++d) part = __builtin_amdgcn_sdot4(fa[d], 0x01010101, part, false);
      part += __shfl_xor(part, 32);
      if (h == 0) rowsum[wave][col] = part;
      bt_v16i acc;
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[t] = 0;
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int a0 = 8 * q + 4 * h;  // this lane's rows of D in register group q
        const bt_v4i rs = *reinterpret_cast<const bt_v4i*>(&rowsum[wave][a0]);
        int tokv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          tokv[t] = acc[4 * q + t] - shift * rs[t] + shift;
          if (8 * q + t + 4 < S) {  // valid for both halves
            hi = max(hi, tokv[t]);
            lo = min(lo, tokv[t]);
          } else if (8 * q + t < S) {
            if (a0 + t < S) {
              hi = max(hi, tokv[t]);
              lo = min(lo, tokv[t]);
            }
          }
        }
        if (r < R) {
          int8_t* dst = trow + a0;
          if (8 * q + 8 <= S) {
            const uint32_t w = pack4(tokv[0], tokv[1], tokv[2], tokv[3]);
            __builtin_memcpy(dst, &w, 4);
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (8 * q + t < S && a0 + t < S) dst[t] = static_cast<int8_t>(tokv[t]);
          }
        }
      }
    }
    const bool bad = __syncthreads_or((hi > 127) | (lo < -128));
    if (threadIdx.x == 0 && overflow && bad) overflow[b] = 1;
  }
}

// One workgroup per (game, mode): L and U cells from one draw each, then P = L @ U.
__global__ __launch_bounds__(kBlock) void sample_basis_kernel(int8_t* P, int8_t* Lo, int8_t* Uo, int64_t B, int S,
                                                              Dist D, uint64_t seed, uint64_t gid0) {
  __shared__ int8_t L[TG_MAX_S * TG_MAX_S], U[TG_MAX_S * TG_MAX_S];
  const uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
  const int cells = S * S, nblk = (cells + 3) >> 2;
  for (int64_t m = blockIdx.x; m < 3 * B; m += gridDim.x) {
    const int64_t b = m / 3;
    const int x = static_cast<int>(m - 3 * b);
    const uint64_t gid = gid0 + static_cast<uint64_t>(b);
    for (int q = threadIdx.x; q < nblk; q += kBlock) {
      const U4 o = philox4x32_10(U4{static_cast<uint32_t>(gid), static_cast<uint32_t>(gid >> 32),
                                    kStreamBasis | static_cast<uint32_t>(x), static_cast<uint32_t>(q)},
                                 k0, k1);
      const uint32_t d[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = 4 * q + t;
        if (c < cells) {
          const int a = c / S, bb = c - a * S;
          const int v = draw_value(d[t], D);
          L[c] = a > bb ? v : (a == bb ? 1 - 2 * static_cast<int>(d[t] & 1u) : 0);
          U[c] = a < bb ? v : (a == bb ? 1 - 2 * static_cast<int>((d[t] >> 1) & 1u) : 0);
        }
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cells; c += kBlock) {
      const int a = c / S, bb = c - a * S;
      int acc = 0;
      const int kmax = a < bb ? a : bb;
      for (int k = 0; k <= kmax; ++k) acc += L[a * S + k] * U[k * S + bb];
      P[m * cells + c] = static_cast<int8_t>(acc);  // |acc| <= S <= 32
      if (Lo) Lo[m * cells + c] = L[c];
      if (Uo) Uo[m * cells + c] = U[c];
    }
    __syncthreads();
  }
}

// The same on the matrix cores, one WAVEFRONT per (game, mode): the cells go into two 32 x 32 byte images in LDS,
// L row-major and U transposed (rows and columns past S are zero), so both fragments of P = L U are one
// ds_read_b128 per lane and the product is ONE v_mfma_i32_32x32x32_i8.  No workgroup barrier: a wavefront owns its
// images.  Same cell -> value rule as sample_basis_kernel (the oracle's).
__global__ __launch_bounds__(kBlock) void sample_basis_mfma_kernel(int8_t* P, int8_t* Lo, int8_t* Uo, int64_t B, int S,
                                                                   Dist D, uint64_t seed, uint64_t gid0) {
  __shared__ __attribute__((aligned(16))) int8_t img[kBlock / 64][2][32 * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int8_t* const Lb = img[wave][0];
  int8_t* const Ut = img[wave][1];
  const uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
  const int cells = S * S, nblk = (cells + 3) >> 2;
  const int col = lane & 31, h = lane >> 5;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (kBlock / 64);
  for (int64_t m = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave; m < 3 * B; m += nwaves) {
    const int64_t b = m / 3;
    const int x = static_cast<int>(m - 3 * b);
    const uint64_t gid = gid0 + static_cast<uint64_t>(b);
    // both images = 2 KiB = 8 dwords per lane
#pragma unroll
    for (int t = 0; t < 8; ++t) reinterpret_cast<uint32_t*>(img[wave])[lane + 64 * t] = 0u;
    for (int q = lane; q < nblk; q += 64) {
      const U4 o = philox4x32_10(U4{static_cast<uint32_t>(gid), static_cast<uint32_t>(gid >> 32),
                                    kStreamBasis | static_cast<uint32_t>(x), static_cast<uint32_t>(q)},
                                 k0, k1);
      const uint32_t d[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = 4 * q + t;
        if (c < cells) {
          const int a = c / S, bb = c - a * S;
          const int v = draw_value(d[t], D);
          Lb[a * 32 + bb] = static_cast<int8_t>(a > bb ? v : (a == bb ? 1 - 2 * static_cast<int>(d[t] & 1u) : 0));
          Ut[bb * 32 + a] = static_cast<int8_t>(a < bb ? v : (a == bb ? 1 - 2 * static_cast<int>((d[t] >> 1) & 1u) : 0));
        }
      }
    }
    // A: row a = col of L (k = 16 h ..), B: column b = col of U = row col of Ut
    const bt_v4i fa = *reinterpret_cast<const bt_v4i*>(Lb + col * 32 + 16 * h);
    const bt_v4i fb = *reinterpret_cast<const bt_v4i*>(Ut + col * 32 + 16 * h);
    bt_v16i acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
    int8_t* Pm = P + m * cells;
    if (col < S) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int a = (t & 3) + 8 * (t >> 2) + 4 * h;  // row of D in register t
        if (a < S) Pm[a * S + col] = static_cast<int8_t>(acc[t]);
      }
    }
    if (Lo || Uo) {
      for (int c = lane; c < cells; c += 64) {
        const int a = c / S, bb = c - a * S;
        if (Lo) Lo[m * cells + c] = Lb[a * 32 + bb];
        if (Uo) Uo[m * cells + c] = Ut[bb * 32 + a];
      }
    }
  }
}

// Change of basis: one workgroup per game, the S^3 int32 tensor lives in LDS ([i][j][k] with the
// k-rows padded to S+1 so that all three fibre directions are bank-conflict free) and is
// transformed IN PLACE one mode at a time; each thread owns whole fibres.
// CB threads per game.  At S = 25 the int32 tensor in LDS (65 KB) allows only two workgroups per CU, and two
// 256-thread workgroups are two wavefronts per SIMD -- nothing to hide the LDS latency of the fibre loop behind:
// 1024 threads there (686 -> 393 us at B = 4096), 256 for the small tensors (S^2 fibres <= 256).
// One game by the CB threads of the calling workgroup; X = the dynamic LDS (see tg_change_basis_i8 for its size).
template <int ST, int CB>
__device__ __forceinline__ void change_basis_game(const int8_t* in, const int32_t* basis, int8_t* out, uint8_t* overflow,
                                                  int64_t b, int Srt, int64_t stride, int* X) {
  const int S = ST ? ST : Srt;
  const int P = S + 1, S2 = S * S, N = S2 * S;
  constexpr int SP = ((ST ? ST : TG_MAX_S) + 3) & ~3;  // fibre length padded to whole ds_read_b128
  const int MP = (S + 3) & ~3;
  int* const Ms = X + ((S2 * P + 3) & ~3);             // this mode's matrix, rows of MP ints
  {
    const int8_t* src = in + b * stride;
    for (int e = threadIdx.x; e < N; e += CB) {
      const int i = e / S2, r = e - i * S2, j = r / S, k = r - j * S;
      X[(i * S + j) * P + k] = src[e];
    }
    __syncthreads();
#pragma unroll 1
    for (int mode = 0; mode < 3; ++mode) {
      const int32_t* M = basis + (b * 3 + mode) * S2;
      // this mode's matrix -> LDS, rows padded to a multiple of 4 entries: the inner loop reads four entries of a
      // row with one broadcast ds_read_b128 (the same address in every lane) instead of one scalar load per MAC
      int mw = 0;
      for (int e = threadIdx.x; e < S * MP; e += CB) {
        const int a = e / MP, t = e - a * MP;
        const int v = t < S ? M[a * S + t] : 0;
        Ms[e] = v;
        mw |= (v + (1 << 23)) >> 24;
      }
      const bool mfits = !__syncthreads_or(mw);  // every entry in [-2^23, 2^23); also: Ms is complete
      // element (f, t) of a fibre: mode 0 walks i, mode 1 walks j, mode 2 walks k
      const int step = mode == 0 ? S * P : (mode == 1 ? P : 1);
      for (int f = threadIdx.x; f < S2; f += CB) {
        const int p = f / S, q = f - p * S;
        const int base = mode == 0 ? p * P + q : (mode == 1 ? p * S * P + q : (p * S + q) * P);
        int x[SP];
        int wide = 0;
#pragma unroll
        for (int t = 0; t < SP; ++t) {
          x[t] = t < S ? X[base + t * step] : 0;
          wide |= (x[t] + (1 << 23)) >> 24;  // non-zero <=> x outside [-2^23, 2^23)
        }
        // int32 * int32 is a quarter-rate multiply; while the matrix entries and this wavefront's values fit 24
        // bits, v_mad_i32_i24 gives the same low 32 bits at full rate
        const bool fast = mfits && __ballot(wide != 0) == 0;
        for (int a = 0; a < S; ++a) {
          const int4* mrow = reinterpret_cast<const int4*>(Ms + a * MP);
          int acc = 0;
#pragma unroll
          for (int t4 = 0; t4 < SP / 4; ++t4) {
            const int4 m4 = mrow[t4];
            if (fast) {
              acc = mad24_pinned(m4.x, x[4 * t4], acc);
              acc = mad24_pinned(m4.y, x[4 * t4 + 1], acc);
              acc = mad24_pinned(m4.z, x[4 * t4 + 2], acc);
              acc = mad24_pinned(m4.w, x[4 * t4 + 3], acc);
            } else {
              acc += m4.x * x[4 * t4] + m4.y * x[4 * t4 + 1] + m4.z * x[4 * t4 + 2] + m4.w * x[4 * t4 + 3];
            }
          }
          X[base + a * step] = acc;
        }
      }
      __syncthreads();
    }
    int8_t* dst = out + b * stride;
    int ovf = 0;
    for (int e = threadIdx.x; e < N; e += CB) {
      const int i = e / S2, r = e - i * S2, j = r / S, k = r - j * S;
      const int v = X[(i * S + j) * P + k];
      ovf |= (v < -128) | (v > 127);
      dst[e] = static_cast<int8_t>(v);
    }
    ovf = __syncthreads_or(ovf);
    if (threadIdx.x == 0 && overflow && ovf) overflow[b] = 1;
  }
}

template <int ST, int CB>
__global__ __launch_bounds__(CB) void change_basis_kernel(const int8_t* in, const int32_t* basis, int8_t* out,
                                                              uint8_t* overflow, int64_t B, int Srt, int64_t stride) {
  extern __shared__ __attribute__((aligned(16))) int X[];
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) change_basis_game<ST, CB>(in, basis, out, overflow, b, Srt, stride, X);
}

// ---------------------------------------------------------------------------------------------------------------
// The same three mode products on the int8 matrix cores (S = 9, 16, 25; round 2).  Per game and mode the product is a
// GEMM whose contracted index has at most 32 values: ONE v_mfma_i32_32x32x32_i8 per 32 rows and operand plane.
//   stage 1 (mode 3):  Y1[(i,j)][c] = sum_k X[(i,j)][k] C[c][k]   tensor rows straight from the state image (k is
//                      contiguous in memory), B operand = rows of C.  |Y1| <= 127 rc may need 16 bits: stored as two
//                      byte planes, lo = low byte (signed), hi = (y + 128) >> 8, in the layout [c][i][j] (j fastest,
//                      32-byte rows) -- the MFMA result has four consecutive rows per register group, so four
//                      consecutive j leave as ONE dword per plane.
//   stage 2 (mode 2):  Y2[(c,i)][b] = sum_j Y1[(c,i)][j] B[b][j]   one tile per c (rows i): two MFMAs (lo, hi planes),
//                      x = (hi << 8) + lo; written IN PLACE over the tile's own block as [c][b][i] (i fastest).
//   stage 3 (mode 1):  T'[a][b][c] = sum_i Y2[(c,b)][i] A[a][i]   one tile per c (rows b); int32 results, range
//                      checked, low bytes into the output image.
// Exact while every basis entry fits int8 and the intermediates fit the two byte planes: max|X| rc <= 32639 and
// max|X| rc rb <= 32639 (a value y is split as lo = signed low byte, hi = (y + 128) >> 8 stored as int8: from y = 32640
// on hi would be 128 and wrap)
// with rc, rb the largest absolute row sums of C and B (checked per game from the data); any other game is done by the
// vector form above (change_basis_game) inside the same launch.
// ---------------------------------------------------------------------------------------------------------------
template <int S>
struct CBGeo {
  static constexpr int N = S * S * S, S2 = S * S;
  static constexpr int NCHUNK = (N + 15) / 16, TAIL = N % 16, IMG = NCHUNK * 16;
  static constexpr int CP = S * 32 + 16;                   // pitch of a c-block: + 16 B spreads the lanes' (= c's) dword writes over 8 banks
  static constexpr int PLANE = S * CP;                     // [c][row][32] bytes
  static constexpr int MAT = 3 * 32 * 32;                  // three int8 matrices, 32 x 32, zero padded
  static constexpr int FAST_BYTES = IMG + 48 + 2 * PLANE + MAT + 64;
  static constexpr int SLOW_BYTES = (((S2 * (S + 1) + 3) & ~3) + S * ((S + 3) & ~3) + 32) * 4;
  static constexpr int LDS_BYTES = FAST_BYTES > SLOW_BYTES ? FAST_BYTES : SLOW_BYTES;
};

// NW wavefronts per workgroup share the game's LDS: the kernel holds one or two workgroups per CU (the vector fallback's
// int32 tensor decides the LDS size), so the wavefronts that hide each other's MFMA / LDS latency must come from the
// workgroup itself.
template <int S, int NW>
__global__ __launch_bounds__(64 * NW) void change_basis_mfma_kernel(const int8_t* in, const int32_t* basis, int8_t* out,
                                                                    uint8_t* overflow, int64_t B, int64_t stride) {
  using G = CBGeo<S>;
  constexpr int kThreads = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) int X[];
  uint8_t* const smem = reinterpret_cast<uint8_t*>(X);
  uint8_t* const img = smem;                                    // the state image, later the output image
  int8_t* const Ylo = reinterpret_cast<int8_t*>(smem + G::IMG + 48);
  int8_t* const Yhi = Ylo + G::PLANE;
  int8_t* const M8 = Yhi + G::PLANE;                            // [mode][row][32]
  int* const red = reinterpret_cast<int*>(M8 + G::MAT);         // [0] not eligible, [1..3] row-sum maxima, [4] overflow, [5] max |X0|
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  constexpr int JP = (S + 3) & ~3;                              // stage 1: rows per i, padded to a multiple of four
  constexpr int NT1 = (S * JP + 31) / 32;                       // stage 1: tiles of 32 rows n' = i JP + j

  for (int64_t g = blockIdx.x; g < B; g += gridDim.x) {
    // ---- 0. state image, int8 matrices (zero padded), eligibility ----
    if (tid < 8) red[tid] = 0;
    for (int e = tid; e < G::MAT / 4; e += kThreads) reinterpret_cast<uint32_t*>(M8)[e] = 0;
    // (the game's three matrices are requested together with its state: one memory round trip, not two)
    constexpr int NBV = (3 * G::S2 + kThreads - 1) / kThreads;
    int bv[NBV];
    {
      const int32_t* Mg = basis + g * 3 * G::S2;
#pragma unroll
      for (int i = 0; i < NBV; ++i) {
        const int e = tid + kThreads * i;
        bv[i] = Mg[e < 3 * G::S2 ? e : 3 * G::S2 - 1];
      }
    }
    {
      const int8_t* src = in + g * stride;
      uint32_t mxe = 0, mxo = 0;  // running max of |x| over even / odd bytes (two 16-bit lanes each)
      for (int c = tid; c < G::NCHUNK; c += kThreads) {
        const uint4 q = load_chunk16<G::TAIL>(src + 16 * c, c == G::NCHUNK - 1);
        *reinterpret_cast<uint4*>(img + 16 * c) = q;
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t s1 = (w[t] >> 7) & 0x01010101u, ax = (w[t] ^ (s1 * 0xFFu)) + s1;  // |x| per byte (128 for -128)
          mxe = pk_max_u16(mxe, ax & 0x00FF00FFu);
          mxo = pk_max_u16(mxo, (ax >> 8) & 0x00FF00FFu);
        }
      }
      const uint32_t m2 = pk_max_u16(mxe, mxo);
      int mine = static_cast<int>(max(m2 & 0xFFFFu, m2 >> 16));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mine = max(mine, __shfl_xor(mine, o));  // one LDS atomic per wavefront, not per lane
      if (lane == 0 && mine) atomicMax(&red[5], mine);
    }
    __syncthreads();
    {
      int bad = 0;
#pragma unroll
      for (int i = 0; i < NBV; ++i) {
        const int e = tid + kThreads * i;
        if (e < 3 * G::S2) {
          const int m = e / G::S2, r = e - m * G::S2, a = r / S, t = r - a * S;
          const int v = bv[i];
          bad |= (v > 127) | (v < -127);
          M8[(m * 32 + a) * 32 + t] = static_cast<int8_t>(v);
        }
      }
      if (bad) red[0] = 1;
    }
    __syncthreads();
    if (tid < 3 * S) {  // row sums of |M| (one row per thread), maxima per mode
      const int m = tid / S, a = tid - m * S;
      int rs = 0;
      for (int t = 0; t < S; ++t) {
        const int v = M8[(m * 32 + a) * 32 + t];
        rs += v < 0 ? -v : v;
      }
      atomicMax(&red[1 + m], rs);
    }
    __syncthreads();
    const int rb = red[2], rc = red[3], mx0 = red[5];  // (the last stage's sums are int32: no condition on A's rows)
    // |Y1| <= mx0 rc and |Y2| <= mx0 rc rb must fit the two byte planes (16 bits); workgroup-uniform
    const bool eligible = red[0] == 0 && static_cast<int64_t>(mx0) * rc <= 32639 && static_cast<int64_t>(mx0) * rc * rb <= 32639;
    if (!eligible) {
      __syncthreads();
      change_basis_game<S, kThreads>(in, basis, out, overflow, g, S, stride, X);
      __syncthreads();
      continue;
    }
    // B-operand fragments of the three stages: row `col` of the mode's matrix, bytes 16h .. 16h+15 (zero beyond S)
    const bt_v4i fC = *reinterpret_cast<const bt_v4i*>(M8 + (2 * 32 + col) * 32 + 16 * h);
    const bt_v4i fB = *reinterpret_cast<const bt_v4i*>(M8 + (1 * 32 + col) * 32 + 16 * h);
    const bt_v4i fA = *reinterpret_cast<const bt_v4i*>(M8 + (0 * 32 + col) * 32 + 16 * h);

    // two byte planes of four consecutive results -> one dword each
    auto split4 = [&](const bt_v16i& acc, int q, uint32_t& lo, uint32_t& hi) {
      lo = pack4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
      hi = pack4((acc[4 * q] + 128) >> 8, (acc[4 * q + 1] + 128) >> 8, (acc[4 * q + 2] + 128) >> 8, (acc[4 * q + 3] + 128) >> 8);
    };

    // ---- 1. contract k with C: rows (i,j) from the image, results to planes [c][i][j] ----
    // The row space is padded to JP = ceil4(S) rows per i (n' = i JP + j): a register group's four consecutive rows then
    // share i and start at a multiple of four, so the four results leave as ONE ALIGNED dword per plane (unaligned LDS
    // dwords stall on gfx950); the JP - S padding rows of every i shadow a valid row and land in the row's padding bytes.
    for (int t1 = wave; t1 < NT1; t1 += NW) {
      const int n = 32 * t1 + col;
      const int ni = n / JP, nj = n - ni * JP;
      const int vi = ni < S ? ni : S - 1, vj = nj < S ? nj : S - 1;  // padding rows shadow a valid one
      const int off = (vi * S + vj) * S + 16 * h;          // 16 bytes of the row (bytes k >= S meet zero rows of C)
      const uint32_t* p4 = reinterpret_cast<const uint32_t*>(img + (off & ~3));
      uint32_t d[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) d[t] = p4[t];
      bt_v4i fx;
#pragma unroll
      for (int t = 0; t < 4; ++t) fx[t] = static_cast<int>(__builtin_amdgcn_alignbyte(d[t + 1], d[t], static_cast<uint32_t>(off & 3)));
      bt_v16i acc;
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[t] = 0;
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fx, fC, acc, 0, 0, 0);  // D[row n'][col c]
      if (col < S) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n0 = 32 * t1 + 8 * q + 4 * h;        // rows n0 .. n0+3 of this register group: one i, j0 % 4 == 0
          const int i0 = n0 / JP, j0 = n0 - i0 * JP;
          if (i0 < S) {
            uint32_t lo, hi;
            split4(acc, q, lo, hi);
            const int o = col * G::CP + i0 * 32 + j0;
            *reinterpret_cast<uint32_t*>(Ylo + o) = lo;
            *reinterpret_cast<uint32_t*>(Yhi + o) = hi;
          }
        }
      }
    }
    __syncthreads();
    // (bytes j >= S of a plane row hold shadows or stale bytes: they meet the zero rows k >= S of the B operand)

    // ---- 2. contract j with B: one tile per c (rows i), in place -> [c][b][i] ----
    for (int c = wave; c < S; c += NW) {
      const int row = col < S ? col : S - 1;             // rows i >= S shadow the last one; never stored
      const bt_v4i xl = *reinterpret_cast<const bt_v4i*>(Ylo + c * G::CP + row * 32 + 16 * h);
      const bt_v4i xh = *reinterpret_cast<const bt_v4i*>(Yhi + c * G::CP + row * 32 + 16 * h);
      bt_v16i al, ah;
#pragma unroll
      for (int t = 0; t < 16; ++t) al[t] = ah[t] = 0;
      al = __builtin_amdgcn_mfma_i32_32x32x32_i8(xl, fB, al, 0, 0, 0);  // D[row i][col b]
      ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(xh, fB, ah, 0, 0, 0);
      // low planes are signed bytes and high planes signed: y = 256 hi + lo
#pragma unroll
      for (int t = 0; t < 16; ++t) al[t] += ah[t] << 8;
      if (col < S) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i0 = 8 * q + 4 * h;
          if (8 * q < S && i0 < S) {                     // (rows i0+t >= S: zero results, written into the row's padding)
            uint32_t lo, hi;
            split4(al, q, lo, hi);
            const int o = c * G::CP + col * 32 + i0;
            *reinterpret_cast<uint32_t*>(Ylo + o) = lo;
            *reinterpret_cast<uint32_t*>(Yhi + o) = hi;
          }
        }
      }
    }
    __syncthreads();

    // ---- 3. contract i with A: one tile per c; the matrix is the A operand (rows a), the tile's rows (c,b) the B
    // operand, so the result has b on the lane and a in the registers: T'[a][b][c] goes to the output image with a lane
    // stride of S bytes (lane = a would stride S^2: every lane on one bank at S = 16) ----
    int hi3 = 0, lo3 = 0;
    for (int c = wave; c < S; c += NW) {
      const int row = col < S ? col : S - 1;
      const bt_v4i xl = *reinterpret_cast<const bt_v4i*>(Ylo + c * G::CP + row * 32 + 16 * h);
      const bt_v4i xh = *reinterpret_cast<const bt_v4i*>(Yhi + c * G::CP + row * 32 + 16 * h);
      bt_v16i al, ah;
#pragma unroll
      for (int t = 0; t < 16; ++t) al[t] = ah[t] = 0;
      al = __builtin_amdgcn_mfma_i32_32x32x32_i8(fA, xl, al, 0, 0, 0);  // D[row a][col b]
      ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(fA, xh, ah, 0, 0, 0);
      if (col < S) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int a = (t & 3) + 8 * (t >> 2) + 4 * h;
          if ((t & 3) + 8 * (t >> 2) < S && a < S) {
            const int v = al[t] + (ah[t] << 8);
            hi3 = max(hi3, v);
            lo3 = min(lo3, v);
            img[(a * S + col) * S + c] = static_cast<uint8_t>(v);
          }
        }
      }
    }
    if (__ballot((hi3 > 127) | (lo3 < -128)) != 0 && lane == 0) red[4] = 1;
    __syncthreads();
    {
      int8_t* dst = out + g * stride;
      for (int c = tid; c < G::NCHUNK; c += kThreads)
        store_chunk16<G::TAIL>(dst + 16 * c, *reinterpret_cast<const uint4*>(img + 16 * c), c == G::NCHUNK - 1);
      if (tid == 0 && overflow && red[4]) overflow[g] = 1;
    }
    __syncthreads();
  }
}

}  // namespace tg

namespace {

int make_dist(const char* fn, const uint32_t* thresholds, const int8_t* values, int nv, tg::Dist* D) {
  if (!thresholds || !values) return tg_internal_fail(TG_ERR_INVALID, "%s: null distribution", fn);
  if (nv < 1 || nv > TG_MAX_VALUES)
    return tg_internal_fail(TG_ERR_INVALID, "%s: n_values=%d outside [1,%d]", fn, nv, TG_MAX_VALUES);
  bool nonzero = false;
  for (int t = 0; t < TG_MAX_VALUES; ++t) D->val[t] = t < nv ? values[t] : 0;
  for (int t = 0; t < TG_MAX_VALUES - 1; ++t) D->thr[t] = t < nv - 1 ? thresholds[t] : 0xffffffffu;
  for (int t = 0; t + 2 < nv; ++t)
    if (thresholds[t] > thresholds[t + 1])
      return tg_internal_fail(TG_ERR_INVALID, "%s: thresholds must be ascending", fn);
  for (int t = 0; t < nv; ++t) {
    const uint64_t lo = t == 0 ? 0 : thresholds[t - 1];
    const uint64_t hi = t == nv - 1 ? (1ull << 32) : thresholds[t];
    if (values[t] != 0 && hi > lo) nonzero = true;
  }
  if (!nonzero) return tg_internal_fail(TG_ERR_INVALID, "%s: distribution never draws a non-zero value", fn);
  D->nv = nv;
  // the 16-bit form of the factor generator: thr16 = ceil(thr / 2^16); always-true thresholds fold into the base
  // value, never-true ones are dropped
  int base = 0;
  D->nthr = 0;
  for (int t = 0; t < TG_MAX_VALUES - 1; ++t) {
    D->thr16[t] = t < nv - 1 ? static_cast<uint32_t>((static_cast<uint64_t>(thresholds[t]) + 0xFFFFu) >> 16) : 0x10000u;
    D->c16[t] = D->delta16[t] = 0;
  }
  for (int t = 0; t < nv - 1; ++t) {
    if (D->thr16[t] == 0) {
      base = t + 1;
    } else if (D->thr16[t] <= 0xFFFFu) {
      const uint32_t c = D->thr16[t] - 1, dl = static_cast<uint16_t>(static_cast<int>(values[t + 1]) - static_cast<int>(values[t]));
      D->c16[D->nthr] = c | (c << 16);
      D->delta16[D->nthr] = dl | (dl << 16);
      ++D->nthr;
    }
  }
  {
    const uint32_t b16 = static_cast<uint16_t>(static_cast<int>(values[base]));
    D->base16 = b16 | (b16 << 16);
  }
  bool nonzero16 = false;  // the same question at 16-bit resolution: is some non-zero value still reachable?
  for (int t = 0; t < nv; ++t) {
    const uint32_t lo = t == 0 ? 0 : D->thr16[t - 1], hi = t == nv - 1 ? 0x10000u : D->thr16[t];
    if (values[t] != 0 && hi > lo) nonzero16 = true;
  }
  if (!nonzero16)
    return tg_internal_fail(TG_ERR_INVALID, "%s: at the generator's 16-bit resolution the distribution never draws a non-zero value", fn);
  return TG_OK;
}

int launched(const char* fn) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tg_internal_fail(TG_ERR_HIP, "%s: %s", fn, hipGetErrorString(e));
  return TG_OK;
}

unsigned grid_for(int64_t blocks) {
  const int64_t cap = 1 << 20;
  return static_cast<unsigned>(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

}  // namespace

extern "C" {

int tg_gen_demos_i8(int8_t* target_out, int8_t* actions_out, uint8_t* overflow, int64_t B, int S,
                    int R, const uint32_t* thresholds, const int8_t* values, int n_values,
                    int shift, uint64_t seed, uint64_t game_id_offset, const int8_t* basis,
                    int64_t game_stride_bytes, tg_stream_t stream) {
  const char* fn = "tg_gen_demos_i8";
  if (B < 0 || S < 1 || S > TG_MAX_S || game_stride_bytes < (int64_t)S * S * S)
    return tg_internal_fail(TG_ERR_INVALID, "%s: bad B/S/stride", fn);
  if (R < 1 || R > 4096) return tg_internal_fail(TG_ERR_INVALID, "%s: R=%d outside [1,4096]", fn, R);
  tg::Dist D;
  if (int rc = make_dist(fn, thresholds, values, n_values, &D)) return rc;
  if (B == 0) return TG_OK;
  if (!target_out || !actions_out) return tg_internal_fail(TG_ERR_INVALID, "%s: null pointer", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // S = 9 / 16 / 25: draw, change of basis, accumulation and both outputs in ONE kernel (tg_genfused.h)
  if (int rc = tg_internal_gen_fused(target_out, actions_out, overflow, basis, B, S, R, D, shift, seed, game_id_offset,
                                     game_stride_bytes, st))
    return rc < 0 ? rc : TG_OK;
  // otherwise: tokens (+ change of basis on the tokens), then the accumulation of tg_gen_from_factors_i8
  const int64_t nvec = B * R * 3;
  const int vec16 = (reinterpret_cast<uintptr_t>(actions_out) & 15) == 0;
  (void)hipGetLastError();
#define TG_GT(ST, M)                                                                                      \
  do {                                                                                                    \
    const int64_t wgs = (nvec + tg::kBlock * M - 1) / (tg::kBlock * M);                                   \
    hipLaunchKernelGGL((tg::gen_tokens_kernel<ST, M>), dim3(grid_for(wgs > 16384 ? 16384 : wgs)),        \
                       dim3(tg::kBlock), 0, st, actions_out, overflow, B, S, R, D, shift, seed,           \
                       game_id_offset, nullptr, vec16);                                                     \
  } while (0)
  switch (S) {
    case 4:  // 24 % of the draws are rejected: several vectors per lane, as long as the chip stays full
      if (nvec >= (int64_t)tg::kBlock * 16 * 2048) TG_GT(4, 16);
      else if (nvec >= (int64_t)tg::kBlock * 4 * 1024) TG_GT(4, 4);
      else TG_GT(4, 1);
      break;
    case 9:  // 4 %
      if (nvec >= (int64_t)tg::kBlock * 4 * 2048) TG_GT(9, 4);
      else TG_GT(9, 1);
      break;
    case 16: TG_GT(16, 1); break;
    case 25: TG_GT(25, 1); break;
    default: TG_GT(0, 1); break;
  }
#undef TG_GT
  if (basis) {
    const size_t lds = 3 * static_cast<size_t>(S) * S * sizeof(int) + 64 * 3 * static_cast<size_t>(S);
    const dim3 bgrid(grid_for(B > 16384 ? 16384 : B)), bblock(192);
    const bool no_mfma = TG_SWITCH("TG_NO_MFMA");  // A/B switch for measurements
    const int mjob = 3 * ((R + 31) / 32);
    const dim3 mgrid(grid_for(B > 65536 ? 65536 : B)), mblock(64 * (mjob < 8 ? mjob : 8));
    switch (no_mfma ? -S : S) {
      case 9: hipLaunchKernelGGL(tg::basis_tokens_mfma_kernel<9>, mgrid, mblock, 0, st, actions_out, overflow, B, R, shift, basis); break;
      case 16: hipLaunchKernelGGL(tg::basis_tokens_mfma_kernel<16>, mgrid, mblock, 0, st, actions_out, overflow, B, R, shift, basis); break;
      case 25: hipLaunchKernelGGL(tg::basis_tokens_mfma_kernel<25>, mgrid, mblock, 0, st, actions_out, overflow, B, R, shift, basis); break;
      case 4: hipLaunchKernelGGL(tg::basis_tokens_kernel<4>, bgrid, bblock, lds, st, actions_out, overflow, B, S, R, shift, basis); break;
      case -9: hipLaunchKernelGGL(tg::basis_tokens_kernel<9>, bgrid, bblock, lds, st, actions_out, overflow, B, S, R, shift, basis); break;
      case -16: hipLaunchKernelGGL(tg::basis_tokens_kernel<16>, bgrid, bblock, lds, st, actions_out, overflow, B, S, R, shift, basis); break;
      case -25: hipLaunchKernelGGL(tg::basis_tokens_kernel<25>, bgrid, bblock, lds, st, actions_out, overflow, B, S, R, shift, basis); break;
      default: hipLaunchKernelGGL(tg::basis_tokens_kernel<0>, bgrid, bblock, lds, st, actions_out, overflow, B, S, R, shift, basis); break;
    }
  }
  if (int rc = launched(fn)) return rc;
  return tg_gen_from_factors_i8(actions_out, target_out, overflow, B, S, R, game_stride_bytes, shift, stream);
}

int tg_sample_basis_i8(int8_t* basis_out, int8_t* lower_out, int8_t* upper_out, int64_t B, int S,
                       const uint32_t* thresholds, const int8_t* values, int n_values,
                       uint64_t seed, uint64_t game_id_offset, tg_stream_t stream) {
  const char* fn = "tg_sample_basis_i8";
  if (B < 0 || S < 1 || S > TG_MAX_S) return tg_internal_fail(TG_ERR_INVALID, "%s: bad B/S", fn);
  tg::Dist D;
  if (int rc = make_dist(fn, thresholds, values, n_values, &D)) return rc;
  // P = L U is narrowed to int8: an entry is a sum of at most S products of two cells, so every value v of
  // the distribution must satisfy S * v^2 <= 127 (with the +-1 diagonals the bound is never exceeded then);
  // beyond that the product could wrap silently and P would not be L U
  for (int t = 0; t < n_values; ++t)
    if (S * static_cast<int>(values[t]) * static_cast<int>(values[t]) > 127)
      return tg_internal_fail(TG_ERR_INVALID, "%s: value %d too large for S=%d (need S*v^2 <= 127: P = L*U must fit int8)",
                              fn, (int)values[t], S);
  if (B == 0) return TG_OK;
  if (!basis_out) return tg_internal_fail(TG_ERR_INVALID, "%s: null pointer", fn);
  const bool no_mfma = TG_SWITCH("TG_NO_MFMA");  // A/B switch for measurements
  (void)hipGetLastError();
  if (!no_mfma)  // one wavefront per matrix, four per workgroup
    hipLaunchKernelGGL(tg::sample_basis_mfma_kernel, dim3(grid_for((3 * B + 3) / 4)), dim3(tg::kBlock), 0,
                       static_cast<hipStream_t>(stream), basis_out, lower_out, upper_out, B, S, D, seed, game_id_offset);
  else
    hipLaunchKernelGGL(tg::sample_basis_kernel, dim3(grid_for(3 * B)), dim3(tg::kBlock), 0,
                       static_cast<hipStream_t>(stream), basis_out, lower_out, upper_out, B, S, D, seed,
                       game_id_offset);
  return launched(fn);
}

int tg_change_basis_i8(const int8_t* state_in, const int32_t* basis, int8_t* state_out,
                       uint8_t* overflow, int64_t B, int S, int64_t game_stride_bytes,
                       tg_stream_t stream) {
  const char* fn = "tg_change_basis_i8";
  if (B < 0 || S < 1 || S > TG_MAX_S || game_stride_bytes < (int64_t)S * S * S)
    return tg_internal_fail(TG_ERR_INVALID, "%s: bad B/S/stride", fn);
  if (B && state_in && state_in == state_out) return tg_internal_fail(TG_ERR_INVALID, "%s: in-place is not supported", fn);
  if (B == 0) return TG_OK;
  if (!state_in || !basis || !state_out) return tg_internal_fail(TG_ERR_INVALID, "%s: null pointer", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // S = 9 / 16 / 25 on aligned layouts: the matrix-core kernel (games it cannot do exactly take the vector form inside it)
  if (!TG_SWITCH("TG_NO_MFMA") && (S == 9 || S == 16 || S == 25) && (reinterpret_cast<uintptr_t>(state_in) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(state_out) & 15) == 0 && game_stride_bytes % 16 == 0) {
    static std::atomic<unsigned> attr_set[3][64];  // per (S, device): the > 64 KiB dynamic LDS opt-in, once
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    const dim3 mgrid(grid_for(B > 4LL * cus ? 4LL * cus : B));
#define TG_CBM(S_, IDX_, NW_)                                                                                  \
  do {                                                                                                         \
    constexpr int ldsb = tg::CBGeo<S_>::LDS_BYTES;                                                             \
    if (ldsb > 64 * 1024 && !attr_set[IDX_][dev].load(std::memory_order_relaxed)) {                            \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tg::change_basis_mfma_kernel<S_, NW_>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);                    \
      if (e != hipSuccess) return tg_internal_fail(TG_ERR_HIP, "%s: %s", fn, hipGetErrorString(e));            \
      attr_set[IDX_][dev].store(1, std::memory_order_relaxed);                                                 \
    }                                                                                                          \
    (void)hipGetLastError();                                                                                   \
    hipLaunchKernelGGL((tg::change_basis_mfma_kernel<S_, NW_>), mgrid, dim3(64 * NW_), ldsb, st, state_in, basis, state_out, \
                       overflow, B, game_stride_bytes);                                                        \
    return launched(fn);                                                                                       \
  } while (0)
    if (S == 9) TG_CBM(9, 0, 4);
    if (S == 16) TG_CBM(16, 1, 4);
    TG_CBM(25, 2, 8);
#undef TG_CBM
  }
  // int32 tensor with rows padded to S+1, then one mode's matrix with rows padded to whole 16-byte reads (+ slack
  // for the run-time-S kernel, whose unrolled row loop may read past the last row into zero-weighted entries)
  const size_t lds = ((static_cast<size_t>(S) * S * (S + 1) + 3) / 4 * 4 + static_cast<size_t>(S) * ((S + 3) / 4 * 4) + 32) * sizeof(int);
  if (lds > 160 * 1024) return tg_internal_fail(TG_ERR_UNSUPPORTED, "%s: S=%d needs %zu B of LDS", fn, S, lds);
  const dim3 grid(grid_for(B));
#define TG_CB(ST, CB)                                                                                  \
  do {                                                                                                 \
    if (lds > 64 * 1024) {                                                                             \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tg::change_basis_kernel<ST, CB>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
      if (e != hipSuccess) return tg_internal_fail(TG_ERR_HIP, "%s: %s", fn, hipGetErrorString(e));    \
    }                                                                                                  \
    (void)hipGetLastError(); hipLaunchKernelGGL((tg::change_basis_kernel<ST, CB>), grid, dim3(CB), lds, st, state_in, basis, state_out,  \
                       overflow, B, S, game_stride_bytes);                                             \
  } while (0)
  switch (S) {
    case 4: TG_CB(4, 256); break;
    case 9: TG_CB(9, 256); break;
    case 16: TG_CB(16, 256); break;
    case 25: TG_CB(25, 1024); break;
    default:
      if (S > 16) TG_CB(0, 1024);
      else TG_CB(0, 256);
      break;
  }
#undef TG_CB
  return launched(fn);
}

}  // extern "C"
