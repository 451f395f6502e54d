// libtensorgame.so -- hand-written gfx950 (MI355X, CDNA4) kernels for the tensor-game hot path
// and the C ABI of include/tensor_game.h.  The single step, expand and the other byte-streaming
// entries are HBM-bound byte work on the vector ALU (state <- state - u(x)v(x)w, zero check); the
// accumulations over many rank-1 terms (generator, long step_many lists) run on the int8 matrix
// cores (tg_mfma.h).  See DESIGN.md.
//
// Kernel families
//   slow_*   : one 256-thread workgroup per game, byte-granular.  Any S <= TG_MAX_S, any alignment.
//   s4_*     : S = 4 in registers only: 4 lanes per game (16 games per wavefront), one dwordx4
//              per lane, tokens as three dwords per lane, ballot nibble for the zero check.
//   s16_step : S = 16 single step, one wavefront per game, registers only.
//   packed_* / rows_* (tg_packed.h, tg_rows.h): aligned layouts, 16-byte chunks, int16 pairs.
//   *_mfma_* (tg_mfma.h): accumulation over many terms on the matrix cores.
//
// The PRODUCT build reads no environment variable and keeps no mutable host state besides per-device
// caches of device constants (atomics): TG_SWITCH() is constant false.  The A/B build (-DTG_AB_SWITCHES)
// turns the TG_* environment switches on.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../../include/tensor_game.h"
#include "tg_device.h"

namespace tg {

#define TG_MAX_ACTIONS 4096  // K / k / R per call

enum Mode { STEP = 0, MANY = 1, EXPAND = 2, GENF = 3 };

// Debug aid: workgroups of the packed/rows kernels that fell back to the exact byte-wise form
// (factors too large for the 16-bit path, or an int8 overflow in step_many).  A silent fallback is
// a 10-50x slowdown, so tests assert that ordinary inputs never take it (tg_debug_fallbacks).
__device__ unsigned long long g_fallback_workgroups = 0;
__device__ __forceinline__ void note_fallback() {
  if (threadIdx.x == 0) atomicAdd(&g_fallback_workgroups, 1ull);
}
// Debug aid: games the matrix-core pass of tg_step_many_i8 could not certify and handed to the lattice kernels
// (each costs a second pass; the reference's {-1,0,1} and the paper's {-2..2} vocabularies should stay at 0).
__device__ unsigned long long g_many_handovers = 0;

struct ApplyArgs {
  const int8_t* in;      // GENF: unused (state starts at zero)
  int8_t* out;
  const int8_t* actions; // (B, nact, 3S)
  uint8_t* done;         // STEP (B) / EXPAND (B,nact)
  int32_t* done_step;    // MANY (B)
  uint8_t* changed;      // EXPAND (B,nact), nullable
  uint8_t* overflow;     // (B) or EXPAND (B,nact), nullable
  int64_t B;
  int64_t in_stride;
  int64_t out_stride;
  int S;
  int nact;
  int shift;
  int only_flagged;      // MANY: redo only the games whose done_step is kNeedsExact (second pass after tg_mfma.h)
  int stream_out;        // EXPAND, S = 4 / 16: the children leave by non-temporal stores (output beyond kStreamOutBytes)
  uint64_t* keys;        // EXPAND (B,nact), nullable: the 64-bit key of every child (tg_expand_keyed_i8)
  int sweep;             // STEP, S = 16 / 25: 1 = the workgroups take the games in reverse order (sweep_index)
};

// Alternating sweeps.  A step kernel streams the whole batch through each XCD's 4 MiB L2; the next launch streams it
// again in the same order, so what the L2 still holds -- the END of the batch -- is evicted before that launch gets
// there: every launch reads everything from beyond L2.  With the direction alternating from launch to launch the tail
// of one sweep is the head of the next, and whatever part of an XCD's share fits its L2 is a hit.  Workgroup b runs on
// XCD b mod 8 (round-robin dispatch), so the order is reversed WITHIN each residue class: a game stays on its XCD.
__device__ __forceinline__ uint32_t sweep_index(uint32_t b, uint32_t n, int reverse) {
  if (!reverse) return b;
  const uint32_t x = b & 7u, t = b >> 3, tx = (n - x + 7u) >> 3;  // tx blocks have residue x
  return ((tx - 1u - t) << 3) | x;
}

// tg_step_i8 at S = 4: from this many bytes of states on a lane awaits its token before it requests its slice
// (s4_step_kernel<.., TW>; placed by tools/step_sizes_bench.py sweeps)
constexpr int64_t kS4TokenWaitBytes = 384ll << 20;

// tg_step_i8 at S = 16 / 25: the state is read by non-temporal loads for footprints in [from, to)
constexpr int64_t kNtLoadsFromBytes = 320ll << 20, kNtLoadsToBytes = 1280ll << 20;

// done_step value by which many_mfma_kernel hands a game to the lattice kernels (never a valid result)
constexpr int32_t kNeedsExact = INT32_MIN;

// =============================================================================================
// slow path: any S, any alignment.  One workgroup per game, one byte per thread-iteration.
// =============================================================================================
// One game (index b) by the whole workgroup.  nzf: TG_MAX_ACTIONS bytes of LDS (MANY only).
template <int MODE>
__device__ __forceinline__ void slow_game(const ApplyArgs& a, int64_t b, uint8_t* nzf) {
  const int S = a.S, S2 = S * S, N = S2 * S, A3 = 3 * S;
  const int tid = threadIdx.x;
  const int8_t* tok = a.actions + b * a.nact * A3;
  if constexpr (MODE == EXPAND) {
    const int8_t* src = a.in + b * a.in_stride;
    for (int c = 0; c < a.nact; ++c) {
      const int8_t* t = tok + c * A3;
      int8_t* dst = a.out + (b * a.nact + c) * a.out_stride;
      int nz = 0, chg = 0, ovf = 0;
      for (int e = tid; e < N; e += kBlock) {
        const int i = e / S2, r = e - i * S2, j = r / S, l = r - j * S;
        const int p = (t[i] - a.shift) * (t[S + j] - a.shift) * (t[2 * S + l] - a.shift);
        const int n = src[e] - p;
        dst[e] = static_cast<int8_t>(n);
        nz |= n & 255;
        chg |= p;
        ovf |= (n + 128);
      }
      nz = __syncthreads_or(nz);
      chg = __syncthreads_or(chg);
      ovf = __syncthreads_or(ovf & ~255);
      if (tid == 0) {
        a.done[b * a.nact + c] = nz ? 0 : 1;
        if (a.changed) a.changed[b * a.nact + c] = chg ? 1 : 0;
        if (a.overflow && ovf) a.overflow[b * a.nact + c] = 1;
      }
    }
  } else {
    if constexpr (MODE == MANY) {
      __syncthreads();
      for (int k = tid; k < a.nact; k += kBlock) nzf[k] = 0;
      __syncthreads();
    }
    const int8_t* src = (MODE == GENF) ? nullptr : a.in + b * a.in_stride;
    int8_t* dst = a.out + b * a.out_stride;
    int nz = 0, ovf = 0;
    for (int e = tid; e < N; e += kBlock) {
      const int i = e / S2, r = e - i * S2, j = r / S, l = r - j * S;
      int acc = (MODE == GENF) ? 0 : src[e];
      for (int k = 0; k < a.nact; ++k) {
        const int8_t* t = tok + k * A3;
        const int p = (t[i] - a.shift) * (t[S + j] - a.shift) * (t[2 * S + l] - a.shift);
        if constexpr (MODE == GENF) {
          acc += p;
        } else {
          acc -= p;
          ovf |= (acc + 128);
          if constexpr (MODE == MANY) {
            if (acc & 255) nzf[k] = 1;
          }
        }
      }
      if constexpr (MODE == GENF) ovf |= (acc + 128);
      dst[e] = static_cast<int8_t>(acc);
      nz |= acc & 255;
    }
    nz = __syncthreads_or(nz);
    ovf = __syncthreads_or(ovf & ~255);
    if (tid == 0) {
      if constexpr (MODE == STEP) a.done[b] = nz ? 0 : 1;
      if constexpr (MODE == MANY) {
        int first = -1;
        for (int k = 0; k < a.nact; ++k)
          if (!nzf[k]) { first = k; break; }
        a.done_step[b] = first;
      }
      if (a.overflow && ovf) a.overflow[b] = 1;
    }
    __syncthreads();
  }
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void slow_kernel(ApplyArgs a) {
  __shared__ __attribute__((aligned(4))) uint8_t nzf[MODE == MANY ? TG_MAX_ACTIONS : 4];
  for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) slow_game<MODE>(a, b, nzf);
}

// =============================================================================================
// helpers of the aligned kernels: 16-byte chunks <-> 32-bit accumulators
// =============================================================================================
__device__ __forceinline__ void unpack16(const uint4& q, int (&acc)[16]) {
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[4 * d + t] = sbyte(w[d], t);
}

// narrow with wrap; nz |= any non-zero output byte; ovf |= bits >= 8 of (n+128) when out of range
__device__ __forceinline__ uint4 pack16(const int (&acc)[16], uint32_t& nz, int& ovf) {
  uint32_t w[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
#pragma unroll
    for (int t = 0; t < 4; ++t) ovf |= acc[4 * d + t] + 128;
    w[d] = pack4(acc[4 * d], acc[4 * d + 1], acc[4 * d + 2], acc[4 * d + 3]);
    nz |= w[d];
  }
  return uint4{w[0], w[1], w[2], w[3]};
}

template <int TAIL>
__device__ __forceinline__ uint4 load_chunk(const int8_t* p, bool tail) {
  if (TAIL != 0 && tail) {  // last chunk of the game: only TAIL bytes belong to it
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < TAIL; ++t) w[t >> 2] |= static_cast<uint32_t>(static_cast<uint8_t>(p[t])) << (8 * (t & 3));
    return uint4{w[0], w[1], w[2], w[3]};
  }
  return *reinterpret_cast<const uint4*>(p);
}

template <int TAIL>
__device__ __forceinline__ void store_chunk(int8_t* p, const uint4& q, bool tail) {
  if (TAIL != 0 && tail) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int t = 0; t < TAIL; ++t) p[t] = static_cast<int8_t>(w[t >> 2] >> (8 * (t & 3)));
    return;
  }
  *reinterpret_cast<uint4*>(p) = q;
}

#include "tg_packed.h"
#include "tg_rows.h"
#include "tg_mfma.h"
#include "tg_genfused.h"

// =============================================================================================
// S = 4 in registers: 4 lanes per game, lane q owns slice i = q (16 bytes = one dwordx4).
// Tokens: 12 bytes per action = three dwords (u | v | w), read by every lane of the game.
// =============================================================================================
struct S4Factors {
  int ui;        // -(u_i) for subtract modes, +u_i for GENF
  int v[4], w[4];
};

template <bool SUB>
__device__ __forceinline__ S4Factors s4_factors(const int* tok3, int q, int shift) {
  const uint32_t du = tok3[0], dv = tok3[1], dw = tok3[2];
  S4Factors f;
  f.ui = __builtin_amdgcn_sbfe(static_cast<int>(du), 8 * q, 8) - shift;
  if constexpr (SUB) f.ui = -f.ui;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f.v[t] = sbyte(dv, t) - shift;
    f.w[t] = sbyte(dw, t) - shift;
  }
  return f;
}

__device__ __forceinline__ void s4_rank1(int (&acc)[16], const S4Factors& f, int& chg) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int uv = mul24_pinned(f.ui, f.v[j]);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int p = __mul24(uv, f.w[l]);
      acc[4 * j + l] += p;
      chg |= p;
    }
  }
}

// One step on one 16-byte slice (S = 4, lane q owns slice i = q), the body of tg_step_i8, of the child-per-team
// tg_expand_i8 and of the streamed stepper.  Packed form: the slice as 8 int16 pairs, 8 saturating v_pk_mad_i16.
// No range check on the factors is needed: with |factor| <= 255 (int8 token, |shift| <= 127, else the 32-bit form)
// u*v is formed exactly and SATURATES beyond int16, and so does (u v) w + x, so every case the 16-bit form cannot
// represent ends outside the int8 range -- exactly the cases where the true result overflows int8 (|x| <= 255 cannot
// bring a saturated product back).  Those lanes redo their slice in 32-bit (wrapped bytes + flag, as the contract
// wants); all others are exact.
// Round 3: the state enters BIASED -- byte b as b + 128 in [0, 255], zero-extended (x ^ 0x80808080, two v_perm_b32) --
// so "the result fits int8" is "the high byte of every int16 result is zero": the range test is an OR of the eight
// results (4 v_or3) instead of eight v_pk_add_u16 + the ORs, at the price of one XOR per output dword: 51 VALU ops per
// lane on the data path instead of 55 (3.58 against 3.69 us per launch at 131 072 games with the one-dword token load
// of s4_step_kernel, tools/s4_share_probe.hip: with 8 wavefronts per SIMD the arithmetic is on the launch's critical
// path).
// nz |= result bytes; ovf |= (n + 128) of the 32-bit form only (test ovf & ~255).
// the slice's 16 bytes as eight pairs of b + 128 (zero-extended): P[2d] = (b0, b1), P[2d+1] = (b2, b3) of dword d
__device__ __forceinline__ void s4_unpack_biased(const uint4& in_slice, uint32_t (&P)[8]) {
  const uint32_t x[4] = {in_slice.x, in_slice.y, in_slice.z, in_slice.w};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t xb = x[d] ^ 0x80808080u;
    P[2 * d] = __builtin_amdgcn_perm(0u, xb, 0x0c010c00u);
    P[2 * d + 1] = __builtin_amdgcn_perm(0u, xb, 0x0c030c02u);
  }
}

// the step on an unpacked slice (P from s4_unpack_biased; in_slice again for the rare 32-bit redo)
__device__ __forceinline__ uint4 s4_step_unpacked(const uint32_t (&P)[8], const uint4& in_slice, uint32_t du, uint32_t dv,
                                                  uint32_t dw, int q, int shift, uint32_t& nz, int& ovf) {
  const uint32_t shp = (static_cast<uint32_t>(shift) & 0xFFFFu) | (static_cast<uint32_t>(shift) << 16);
  const int ui = shift - __builtin_amdgcn_sbfe(static_cast<int>(du), 8 * q, 8);  // -(u_i)
  const uint32_t uip = __builtin_amdgcn_perm(static_cast<uint32_t>(ui), static_cast<uint32_t>(ui), 0x05040100u);
  const uint32_t yv = dv << 8, yw = dw << 8;
  const uint32_t vA = pk_sub_i16(__builtin_amdgcn_perm(dv, yv, 0x0A050804u), shp);  // (v0, v1)
  const uint32_t vB = pk_sub_i16(__builtin_amdgcn_perm(dv, yv, 0x0B070906u), shp);  // (v2, v3)
  const uint32_t wA = pk_sub_i16(__builtin_amdgcn_perm(dw, yw, 0x0A050804u), shp);  // (w0, w1)
  const uint32_t wB = pk_sub_i16(__builtin_amdgcn_perm(dw, yw, 0x0B070906u), shp);  // (w2, w3)
  const uint32_t uvA = pk_mad_i16_sat(vA, uip, 0u), uvB = pk_mad_i16_sat(vB, uip, 0u);
  uint32_t A[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {  // row j = d: -u v_j is the low (j even) or high (j odd) half of uvA / uvB
    const uint32_t uv = d < 2 ? uvA : uvB;
    if (d & 1) {
      A[2 * d] = pk_mad_i16_sat_hi(uv, wA, P[2 * d]);
      A[2 * d + 1] = pk_mad_i16_sat_hi(uv, wB, P[2 * d + 1]);
    } else {
      A[2 * d] = pk_mad_i16_sat_lo(uv, wA, P[2 * d]);
      A[2 * d + 1] = pk_mad_i16_sat_lo(uv, wB, P[2 * d + 1]);
    }
  }
  uint32_t w[4], ovf16 = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    ovf16 |= A[2 * d] | A[2 * d + 1];
    w[d] = __builtin_amdgcn_perm(A[2 * d + 1], A[2 * d], 0x06040200u) ^ 0x80808080u;
    nz |= w[d];
  }
  uint4 pk{w[0], w[1], w[2], w[3]};
  const bool wide_shift = static_cast<unsigned>(shift + 127) > 254u;  // uniform; factors may exceed 255
  if (__builtin_expect(wide_shift || (ovf16 & 0xFF00FF00u), 0)) {  // rare, per lane: exact 32-bit form of this slice
    const int cur[3] = {static_cast<int>(du), static_cast<int>(dv), static_cast<int>(dw)};
    const S4Factors f = s4_factors<true>(cur, q, shift);
    int acc[16], chg = 0;
    nz = 0;
    unpack16(in_slice, acc);
    s4_rank1(acc, f, chg);
    pk = pack16(acc, nz, ovf);
  }
  return pk;
}

__device__ __forceinline__ uint4 s4_step_slice(const uint4 in_slice, uint32_t du, uint32_t dv, uint32_t dw, int q,
                                               int shift, uint32_t& nz, int& ovf) {
  uint32_t P[8];
  s4_unpack_biased(in_slice, P);
  return s4_step_unpacked(P, in_slice, du, dv, dw, q, shift, nz, ovf);
}

// Digit form of the same step (round 3, second half): a dword of the slice -- row j, elements l = 0..3 -- is read as ONE
// base-256 integer whose digits are the biased bytes b_l = x_l + 128, and the game's w as the integer
// W = sum_l w_l 256^l (= the token dword minus shift * 0x01010101: tokens below 128 make that exact).  The update of
// the whole row is then linear in ONE 32-bit multiply-add,
//     X'_j = X_j + v_j * G  (mod 2^32),   G = -u_i * W,
// and X'_j is the packed result exactly when every digit b_l - u_i v_j w_l stays in [0, 255] (no carry or borrow
// crosses a byte).  That is guaranteed up front, not checked afterwards: all twelve token bytes <= 3 and
// 0 <= shift <= 3 bound every factor by F = max(shift, 3 - shift) <= 3, and the slice's L1 norm (four v_sad_u8 on the
// biased dwords, which need no unpacking either) bounds every |x_l|; L1 <= 127 - F^3 keeps all results inside int8.
// 3 VALU per dword (bias, v_mad_u64_u32, unbias) + 1 for the norm instead of 9 for unpack / two packed MADs / pack /
// range: ~31 instead of ~51 on the data path.  Lanes outside the guarantee (tokens of a wider vocabulary, large
// entries, other shifts) take s4_step_unpacked; results are identical wherever both apply (tests force each form).
// pre: the part that needs the state only (runs while the token dword is still on its way)
__device__ __forceinline__ uint32_t s4_digits_pre(const uint4& in_slice, uint32_t (&xb)[4]) {
  const uint32_t x[4] = {in_slice.x, in_slice.y, in_slice.z, in_slice.w};
  uint32_t l1 = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    xb[d] = x[d] ^ 0x80808080u;
    l1 = __builtin_amdgcn_sad_u8(xb[d], 0x80808080u, l1);
  }
  return l1;
}
// wave-uniform: the largest slice norm the digit form accepts under this shift, -1 when it never applies
// a * b mod 2^32 by v_mad_u64_u32 (full rate on gfx950: 4.9 issue cycles; hipcc's v_mul_lo_u32 is a quarter-rate instruction)
__device__ __forceinline__ uint32_t mul_lo_mad(uint32_t x, uint32_t y) {
  uint64_t r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(x), "v"(y) : "vcc");
  return static_cast<uint32_t>(r);
}

__host__ __device__ __forceinline__ int s4_digits_limit(int shift) {
  const int F = shift > 3 - shift ? shift : 3 - shift;
  return static_cast<unsigned>(shift) <= 3u ? 127 - F * F * F : -1;
}
// returns false when this lane must take the packed form; nz |= result bytes
__device__ __forceinline__ bool s4_step_digits(const uint32_t (&xb)[4], uint32_t l1, int limit, uint32_t du, uint32_t dv,
                                               uint32_t dw, int q, int shift, uint4& out, uint32_t& nz) {
  const uint32_t wide = (du | dv | dw) & 0xFCFCFCFCu;
  // -(u_i): byte q of du comes down by v_alignbyte_b32 (shifts by q BYTES: no 8 * q), then one SDWA subtract
  const uint32_t nui = static_cast<uint32_t>(shift) - (__builtin_amdgcn_alignbyte(du, du, static_cast<uint32_t>(q)) & 255u);
  const uint32_t W = dw - static_cast<uint32_t>(shift) * 0x01010101u;
  const uint32_t G = mul_lo_mad(nui, W);
  uint32_t o[4], vj[4];
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(vj[0]) : "v"(dv), "s"(shift));
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(vj[1]) : "v"(dv), "s"(shift));
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(vj[2]) : "v"(dv), "s"(shift));
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(vj[3]) : "v"(dv), "s"(shift));
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = (xb[d] + vj[d] * G) ^ 0x80808080u;
  out = uint4{o[0], o[1], o[2], o[3]};
  nz |= o[0] | o[1] | o[2] | o[3];
  return wide == 0 && static_cast<int>(l1) <= limit;
}

// One step on one slice, digit form first and the packed form for the lanes it does not cover (the body shared by
// tg_expand_i8's child teams and the streamed stepper; s4_step_kernel spells the two halves out around its loads).
__device__ __forceinline__ uint4 s4_step_tiered(const uint4& in_slice, uint32_t du, uint32_t dv, uint32_t dw, int q, int shift,
                                                int digits_limit, uint32_t& nz, int& ovf) {
  uint32_t xb[4];
  const uint32_t l1 = s4_digits_pre(in_slice, xb);
  uint4 o;
  uint32_t dnz = 0;
  if (__builtin_expect(s4_step_digits(xb, l1, digits_limit, du, dv, dw, q, shift, o, dnz), 1)) {
    nz |= dnz;
    return o;
  }
  return s4_step_slice(in_slice, du, dv, dw, q, shift, nz, ovf);
}

// The resident stepper's form of the same step: the slice stays BIASED (x ^ 0x80808080) between steps and its L1 norm is
// carried -- the norm of the new slice is at once this step's zero test (l1 == 0) and the next step's precondition, so a step
// is four multiply-adds and four v_sad_u8 (s4_step_tiered: four xor in, four v_sad_u8, four multiply-adds, four xor out,
// three or).  A lane the digit form does not cover un-biases, takes s4_step_slice and biases again.
__device__ __forceinline__ void s4_step_biased(uint4& xb, uint32_t& l1, uint32_t du, uint32_t dv, uint32_t dw, int q, int shift,
                                               int digits_limit, int& ovf) {
  constexpr uint32_t BIAS = 0x80808080u;
  const uint32_t wide = (du | dv | dw) & 0xFCFCFCFCu;
  const uint32_t nui = static_cast<uint32_t>(shift) - (__builtin_amdgcn_alignbyte(du, du, static_cast<uint32_t>(q)) & 255u);
  const uint32_t W = dw - static_cast<uint32_t>(shift) * 0x01010101u;
  const uint32_t G = mul_lo_mad(nui, W);
  uint32_t vj[4];
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(vj[0]) : "v"(dv), "s"(shift));
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(vj[1]) : "v"(dv), "s"(shift));
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(vj[2]) : "v"(dv), "s"(shift));
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(vj[3]) : "v"(dv), "s"(shift));
  uint4 o{xb.x + vj[0] * G, xb.y + vj[1] * G, xb.z + vj[2] * G, xb.w + vj[3] * G};
  if (__builtin_expect(!(wide == 0 && static_cast<int>(l1) <= digits_limit), 0)) {
    uint32_t nz = 0;
    const uint4 r = s4_step_slice(uint4{xb.x ^ BIAS, xb.y ^ BIAS, xb.z ^ BIAS, xb.w ^ BIAS}, du, dv, dw, q, shift, nz, ovf);
    o = uint4{r.x ^ BIAS, r.y ^ BIAS, r.z ^ BIAS, r.w ^ BIAS};
  }
  xb = o;
  l1 = __builtin_amdgcn_sad_u8(o.w, BIAS, __builtin_amdgcn_sad_u8(o.z, BIAS, __builtin_amdgcn_sad_u8(o.y, BIAS, __builtin_amdgcn_sad_u8(o.x, BIAS, 0u))));
}

// Reductions over the four lanes of a team (a DPP quad): two VALU instructions with the exchange folded in (v_add_u32_dpp /
// v_or_b32_dpp), every lane ends with the team's value.  team_any<4> does the same through a ballot: v_cmp + four v_and +
// two 64-bit compares per use -- a third of the resident stepper's step before round 4.
__device__ __forceinline__ uint32_t quad_sum(uint32_t x) {
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
  return x;
}
__device__ __forceinline__ uint32_t quad_or(uint32_t x) {
  x |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0xB1, 0xf, 0xf, false));
  x |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x4E, 0xf, 0xf, false));
  return x;
}

// s4_step_biased for a caller that has tested the tokens of a whole BLOCK of steps at once (`wide`: some token byte of the
// block exceeds 3, team-uniform): the per-step or / and / compare of the twelve bytes leaves the step.
__device__ __forceinline__ void s4_step_biased_blk(uint4& xb, uint32_t& l1, uint32_t du, uint32_t dv, uint32_t dw, int q, int shift,
                                                   int digits_limit, bool wide, int& ovf) {
  constexpr uint32_t BIAS = 0x80808080u;
  const uint32_t nui = static_cast<uint32_t>(shift) - (__builtin_amdgcn_alignbyte(du, du, static_cast<uint32_t>(q)) & 255u);
  const uint32_t W = dw - static_cast<uint32_t>(shift) * 0x01010101u;
  const uint32_t G = mul_lo_mad(nui, W);
  uint32_t vj[4];
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(vj[0]) : "v"(dv), "s"(shift));
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(vj[1]) : "v"(dv), "s"(shift));
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(vj[2]) : "v"(dv), "s"(shift));
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(vj[3]) : "v"(dv), "s"(shift));
  uint4 o{xb.x + vj[0] * G, xb.y + vj[1] * G, xb.z + vj[2] * G, xb.w + vj[3] * G};
  if (__builtin_expect(wide || static_cast<int>(l1) > digits_limit, 0)) {
    uint32_t nz = 0;
    const uint4 r = s4_step_slice(uint4{xb.x ^ BIAS, xb.y ^ BIAS, xb.z ^ BIAS, xb.w ^ BIAS}, du, dv, dw, q, shift, nz, ovf);
    o = uint4{r.x ^ BIAS, r.y ^ BIAS, r.z ^ BIAS, r.w ^ BIAS};
  }
  xb = o;
  l1 = __builtin_amdgcn_sad_u8(o.w, BIAS, __builtin_amdgcn_sad_u8(o.z, BIAS, __builtin_amdgcn_sad_u8(o.y, BIAS, __builtin_amdgcn_sad_u8(o.x, BIAS, 0u))));
}

// The game's 12 token bytes as three dwords (u | v | w) in every lane of its 4-lane team from ONE dword load per lane:
// lane q loads dword min(q, 2) and the team exchanges them by DPP quad broadcasts (three v_mov_b32_dpp).  A
// global_load_dwordx3 per lane asks the memory pipeline for 48 bytes per game where 12 are distinct; with the token
// buffers of a rollout coming from beyond L2 that is 0.06 us of a 3.6 us launch at 131 072 games.
// blk_tok: the (wave-uniform) token base of the workgroup; team: the game's index within the workgroup.
__device__ __forceinline__ uint32_t s4_team_token_load(const int8_t* blk_tok, int team, int q) {
  const uint32_t off = __umul24(static_cast<uint32_t>(team), 12u) + 4u * static_cast<uint32_t>(q < 3 ? q : 2);  // scalar base + 32-bit lane offset
  return *reinterpret_cast<const uint32_t*>(blk_tok + off);
}
__device__ __forceinline__ void s4_team_token_bcast(uint32_t mine, uint32_t& du, uint32_t& dv, uint32_t& dw) {
  du = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(mine), 0x00, 0xf, 0xf, true));  // quad_perm [0,0,0,0]
  dv = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(mine), 0x55, 0xf, 0xf, true));  // [1,1,1,1]
  dw = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(mine), 0xAA, 0xf, 0xf, true));  // [2,2,2,2]
}
__device__ __forceinline__ void s4_team_tokens(const int8_t* blk_tok, int team, int q, uint32_t& du, uint32_t& dv, uint32_t& dw) {
  s4_team_token_bcast(s4_team_token_load(blk_tok, team, q), du, dv, dw);
}

// =============================================================================================
// tg_step_i8 at S = 4: the single step (round 3; s4_kernel below keeps step_many, gen_from_factors and the
// team-per-parent expand).  4 lanes per game, 16 games per wavefront; one token dword and one 16-byte slice per lane;
// no LDS, no barrier.  Everything that depends on blockIdx is SCALAR 64-bit math; the per-lane part is a 32-bit offset
// (host guarantees strides < 2^20).
//   NTL: the state is read by non-temporal loads (batches beyond the caches: the lines a launch reads are not worth a
//        place in the Infinity Cache when the next launch's reads evict them anyway);
//   TW:  a lane waits for its token before it requests its slice.  For batches far beyond the Infinity Cache only: the
//        wait halves the state requests a wavefront keeps in flight, and the HBM side serves the thinner stream better
//        (2 GiB of states: 815 -> 793 us, 512 MiB: 202 -> 188; at 256 MiB the same wait costs 8 %).
// =============================================================================================
// Its own slim argument block (64 bytes: two s_load_dwordx8, one scalar-load round trip) and 32-bit strides: with 8 wavefronts per SIMD
// every instruction in front of the loads, and every VALU instruction behind them, is on the launch's critical path
// (one VALU instruction per lane = 0.014 us of a 3.6 us launch at 131 072 games).
struct S4StepArgs {
  const int8_t* in;
  int8_t* out;
  const int8_t* actions;
  uint8_t* done;
  uint8_t* overflow;
  int64_t B;
  uint32_t stride;  // in == out stride (tg_step_i8 has one), < 2^20
  int shift;
  int digits_limit;  // s4_digits_limit(shift), from the host (a branch in front of the loads' consumers costs a block)
  int sweep;         // 1 = the workgroups take the games in reverse order (sweep_index); still 64 bytes of arguments
};

//   DIG: the digit form (s4_step_digits) first, the packed form for the lanes it does not cover; false only in the
//        A/B library (TG_S4_NO_DIGITS), for tests and measurements of the packed form alone.
template <bool NTL, bool TW, bool DIG = true>
__global__ __launch_bounds__(kBlock) void s4_step_kernel(S4StepArgs a) {
  constexpr int GPB = kBlock / 4;  // 64 games per workgroup
  const int64_t g0 = static_cast<int64_t>(sweep_index(blockIdx.x, gridDim.x, a.sweep)) * GPB;
  const int nlive = static_cast<int>(min(static_cast<int64_t>(GPB), a.B - g0));
  const int lg_raw = threadIdx.x >> 2, q = threadIdx.x & 3;
  const bool live = lg_raw < nlive;
  const int lg = live ? lg_raw : nlive - 1;  // dead lanes shadow the last live game, stores predicated off
  const uint32_t off = __umul24(static_cast<uint32_t>(lg), a.stride) + 16u * q;
  const int8_t* const in_blk = a.in + g0 * a.stride;
  uint32_t du, dv, dw;
  uint4 pk;
  auto load_state = [&]() {
    if constexpr (NTL) {
      const v4u_t v = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(in_blk + off));
      pk = uint4{v.x, v.y, v.z, v.w};
    } else {
      pk = *reinterpret_cast<const uint4*>(in_blk + off);
    }
  };
  uint32_t P[8], xb[4], l1 = 0;
  auto state_only = [&]() {  // what can be done before the token is there
    if constexpr (DIG) l1 = s4_digits_pre(pk, xb);
    else s4_unpack_biased(pk, P);
  };
  if constexpr (TW) {  // token, wait, slice (the throttled order for batches far beyond the caches)
    s4_team_tokens(a.actions + g0 * 12, lg, q, du, dv, dw);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    load_state();
    state_only();
  } else {
    // The slice FIRST, then the token dword: the states of an in-place rollout sit in L2 / the Infinity Cache, the
    // step's token block comes from wherever its producer left it -- so the slice is biased and measured (or
    // unpacked) while the token is still on its way (vmcnt retires in order).
    // Both requests and both waits are written out: hipcc otherwise issues the token load BEHIND the wait for the slice
    // (a sched_barrier does not hold it: the load is placed at instruction selection), which puts two memory round
    // trips in series.  The "+v" operands of the waits tie the consumers of each register to its wait.
    v4u_t sv;
    uint32_t mine;
    const int8_t* const tok_blk = a.actions + g0 * 12;
    const uint32_t toff = __umul24(static_cast<uint32_t>(lg), 12u) + min(4u * static_cast<uint32_t>(q), 8u);
    if constexpr (NTL) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(sv) : "v"(off), "s"(in_blk) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(sv) : "v"(off), "s"(in_blk) : "memory");
    asm volatile("global_load_dword %0, %1, %2" : "=&v"(mine) : "v"(toff), "s"(tok_blk) : "memory");
    asm volatile("s_waitcnt vmcnt(1)" : "+v"(sv) : : "memory");
    pk = uint4{sv.x, sv.y, sv.z, sv.w};
    state_only();
    if constexpr (DIG)  // (l1 / P as operands: the state-only work stays in front of this wait)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(mine), "+v"(l1) : : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(mine), "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+v"(P[5]), "+v"(P[6]), "+v"(P[7]) : : "memory");
    s4_team_token_bcast(mine, du, dv, dw);
  }
  uint32_t nz = 0;
  int ovf = 0;
  if constexpr (DIG) {
    uint4 o;
    if (__builtin_expect(s4_step_digits(xb, l1, a.digits_limit, du, dv, dw, q, a.shift, o, nz), 1)) {
      pk = o;
    } else {
      nz = 0;
      pk = s4_step_slice(pk, du, dv, dw, q, a.shift, nz, ovf);
    }
  } else {
    pk = s4_step_unpacked(P, pk, du, dv, dw, q, a.shift, nz, ovf);
  }
  // (skipping the store of untouched slices, as the S >= 9 kernels do in place, is SLOWER here: 16-byte holes inside
  // 64-byte games turn full-line writes into partial ones -- 2.83 -> 3.05 us at BASELINE config 2)
  if (live) *reinterpret_cast<uint4*>(a.out + g0 * a.stride + off) = pk;
  // done: the OR of the team's four slices by two quad-permuting DPP ORs (every lane is active here; dead lanes hold
  // a copy of the last live game), then one byte per game through the workgroup's scalar base + a 32-bit lane offset
  // (the ballot form cost 8 VALU instructions on the q == 0 lanes, this costs 4 on all)
  uint32_t t1, t2;
  asm("s_nop 1\n\tv_or_b32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(t1) : "v"(nz));
  asm("s_nop 1\n\tv_or_b32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(t2) : "v"(t1));
  uint8_t* const done_blk = a.done + g0;
  if (q == 0 && live) done_blk[static_cast<uint32_t>(lg)] = t2 ? 0 : 1;
  // the flag is sticky and only ever set to 1: a lane whose slice overflowed stores it itself (rare), so the common
  // path carries no team reduction for it
  if (__builtin_expect((ovf & ~255) != 0, 0) && a.overflow && live) (a.overflow + g0)[lg] = 1;
}

// =============================================================================================
// tg_step_emit at S = 4 ("a fused step + emit model input kernel removes a full extra pass over the
// state"): one env step on the history ring AND the (B,T,4,4,4) float model input of the new state in one launch.
// The step writes the new head into ring slot (head+1) mod T; frame 0 of the output comes from the registers that hold
// it, frame 1 (the old head) from the registers the step read it into, older frames from the ring.
// Lane mapping: 4 lanes per game as in s4_step_kernel, but TRANSPOSED -- lane q holds dword q of every slice, i.e. the
// elements (i = d, j = q, l = 0..3) for d = 0..3 -- so that the four conversions of dword d leave as 16 contiguous
// output bytes per lane and 64 contiguous bytes per team (float32; 32 bytes for the 16-bit types): whole sectors per
// store instruction instead of 16-byte pieces 64 bytes apart.  The arithmetic is s4_step_slice's with the roles of u
// and v exchanged (the lane constant is -v_q, the dword index selects u_d).
// =============================================================================================
struct StepEmitArgs {
  int8_t* ring;
  const int8_t* actions;
  void* out;
  float* scalars;
  uint8_t* done;
  uint8_t* overflow;
  int64_t B;
  int64_t frame_stride;
  int64_t game_stride;
  int T;
  int head;
  int shift;
  float t_step;
};

// four int8 -> four float32 at dst (16 bytes); NT: non-temporal store
template <bool NT>
__device__ __forceinline__ void s4_emit_f32(float* dst, uint32_t w) {
  const uint4 o{__float_as_uint(static_cast<float>(sbyte(w, 0))), __float_as_uint(static_cast<float>(sbyte(w, 1))),
                __float_as_uint(static_cast<float>(sbyte(w, 2))), __float_as_uint(static_cast<float>(sbyte(w, 3)))};
  if constexpr (NT) store16_nt(dst, o);
  else *reinterpret_cast<uint4*>(dst) = o;
}
// four int8 -> two dwords of two 16-bit floats each
template <typename OutT>
__device__ __forceinline__ uint2 s4_cvt16(uint32_t w) {
  const float f0 = static_cast<float>(sbyte(w, 0)), f1 = static_cast<float>(sbyte(w, 1)), f2 = static_cast<float>(sbyte(w, 2)),
              f3 = static_cast<float>(sbyte(w, 3));
  uint32_t lo, hi;
  if constexpr (std::is_same<OutT, __half>::value) {
    typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t x = __builtin_amdgcn_cvt_pkrtz(f0, f1), y = __builtin_amdgcn_cvt_pkrtz(f2, f3);  // |x| <= 128: exact
    __builtin_memcpy(&lo, &x, 4);
    __builtin_memcpy(&hi, &y, 4);
  } else {  // bfloat16 = the upper half of the float32 (at most 8 significant bits: exact)
    lo = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
    hi = __builtin_amdgcn_perm(__float_as_uint(f3), __float_as_uint(f2), 0x07060302u);
  }
  return uint2{lo, hi};
}
// One frame (the lane's dwords y[d] = elements (i = d, j = q, l = 0..3)) -> the output, `frame` = element (0, 0, 0) of
// this game's frame.  float32: dword d of lane q is 16 output bytes at element 16 d + 4 q -- 64 contiguous bytes per
// team and instruction.  16-bit types: a dword is only 8 output bytes, so the lanes of a PAIR (q, q ^ 1) exchange
// dwords (DPP quad_perm [1,0,3,2]) and the even lane stores rows d = 0, 2, the odd lane rows d = 1, 3, each as 16
// bytes covering (j = 2 p, 2 p + 1): again 16-byte stores and 64 contiguous bytes per team and instruction (8-byte
// stores ran the 2^20-game case at 0.65 of step + emit_frames).
template <typename OutT, bool NT>
__device__ __forceinline__ void s4_emit_frame(OutT* frame, const uint32_t (&y)[4], int q) {
  if constexpr (sizeof(OutT) == 4) {
#pragma unroll
    for (int d = 0; d < 4; ++d) s4_emit_f32<NT>(frame + 16 * d + 4 * q, y[d]);
  } else {
    const bool odd = (q & 1) != 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // rows d = 2 h (even lane) / 2 h + 1 (odd lane)
      const uint32_t give = odd ? y[2 * h] : y[2 * h + 1];   // what the partner stores of mine
      const uint32_t got = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(give), 0xB1, 0xf, 0xf, true));
      const uint32_t mine = odd ? y[2 * h + 1] : y[2 * h];
      const uint2 a = s4_cvt16<OutT>(odd ? got : mine), b = s4_cvt16<OutT>(odd ? mine : got);  // (j = 2p, j = 2p + 1)
      const uint4 o{a.x, a.y, b.x, b.y};
      OutT* const dst = frame + 16 * (2 * h + (odd ? 1 : 0)) + 4 * (q & ~1);
      if constexpr (NT) store16_nt(dst, o);
      else *reinterpret_cast<uint4*>(dst) = o;
    }
  }
}

template <typename OutT, bool NT>
__global__ __launch_bounds__(kBlock) void s4_step_emit_kernel(StepEmitArgs a) {
  constexpr int GPB = kBlock / 4;
  const int64_t g0 = static_cast<int64_t>(blockIdx.x) * GPB;
  const int nlive = static_cast<int>(min(static_cast<int64_t>(GPB), a.B - g0));
  const int lg_raw = threadIdx.x >> 2, q = threadIdx.x & 3;
  const bool live = lg_raw < nlive;
  const int lg = live ? lg_raw : nlive - 1;
  uint32_t du, dv, dw;
  s4_team_tokens(a.actions + g0 * 12, lg, q, du, dv, dw);
  const int64_t g = g0 + lg;
  int8_t* const game = a.ring + g * a.game_stride;
  const int nxt = a.head + 1 < a.T ? a.head + 1 : 0;
  const int8_t* const src = game + a.head * a.frame_stride + 4 * q;
  uint32_t x[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) x[d] = *reinterpret_cast<const uint32_t*>(src + 16 * d);  // (i = d, j = q, l = 0..3)
  uint32_t nz = 0;
  int ovf = 0;
  // u and v exchanged: the lane's constant is -v_q, dword d takes u_d
  const uint4 nw = s4_step_tiered(uint4{x[0], x[1], x[2], x[3]}, dv, du, dw, q, a.shift, s4_digits_limit(a.shift), nz, ovf);
  const uint32_t y[4] = {nw.x, nw.y, nw.z, nw.w};
  // (`live` is uniform over a team, so the pair exchange of the 16-bit path stays inside the active lanes)
  OutT* const out = static_cast<OutT*>(a.out) + g * (a.T * 64);
  if (live) {
    int8_t* const dst = game + nxt * a.frame_stride + 4 * q;
#pragma unroll
    for (int d = 0; d < 4; ++d) *reinterpret_cast<uint32_t*>(dst + 16 * d) = y[d];
    s4_emit_frame<OutT, NT>(out, y, q);                    // frame 0: the new head
    if (a.T > 1) s4_emit_frame<OutT, NT>(out + 64, x, q);  // frame 1: the old head
    int slot = a.head;
    for (int f = 2; f < a.T; ++f) {                        // older frames from the ring
      slot = slot > 0 ? slot - 1 : a.T - 1;
      const int8_t* const old = game + slot * a.frame_stride + 4 * q;
      uint32_t z[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) z[d] = *reinterpret_cast<const uint32_t*>(old + 16 * d);
      s4_emit_frame<OutT, NT>(out + 64 * f, z, q);
    }
  }
  const bool any_nz = team_any<4>(nz != 0);
  if (q == 0 && live) {
    a.done[g] = any_nz ? 0 : 1;
    if (a.scalars) a.scalars[g] = a.t_step;
  }
  if (__builtin_expect((ovf & ~255) != 0, 0) && a.overflow && live) a.overflow[g] = 1;
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void s4_kernel(ApplyArgs a) {
  // Addressing: everything that depends on blockIdx is SCALAR 64-bit math (SALU); the per-lane
  // part is a small 32-bit offset (host guarantees strides < 2^20).  At the BASELINE cfg2 shape
  // (65 536 games, ~2.5 us per launch) vector 64-bit multiplies were 0.5 us of the launch.
  constexpr int GPB = kBlock / 4;  // 64 games per workgroup
  const int64_t g0 = static_cast<int64_t>(blockIdx.x) * GPB;
  const int nlive = static_cast<int>(min(static_cast<int64_t>(GPB), a.B - g0));
  const int lg_raw = threadIdx.x >> 2, q = threadIdx.x & 3;
  const bool live = lg_raw < nlive;
  const int lg = live ? lg_raw : nlive - 1;  // dead lanes shadow the last live game, stores predicated off
  static_assert(MODE == MANY || MODE == GENF || MODE == EXPAND, "the single step is s4_step_kernel");
  const int nact = a.nact;
  const int* tok = reinterpret_cast<const int*>(a.actions + g0 * nact * 12) + lg * nact * 3;
  const int8_t* in_blk = a.in + g0 * a.in_stride;
  const uint32_t in_off = __umul24(lg, static_cast<uint32_t>(a.in_stride)) + 16u * q;
  uint4 pk{0, 0, 0, 0};
  if constexpr (MODE != GENF) pk = *reinterpret_cast<const uint4*>(in_blk + in_off);
  int ovf = 0;

  if constexpr (MODE == MANY || MODE == GENF) {
    int8_t* out_blk = a.out + g0 * a.out_stride;
    const uint32_t out_off = __umul24(lg, static_cast<uint32_t>(a.out_stride)) + 16u * q;
    {
      // exact 32-bit form: GENF always; MANY for teams the lattice form below hands over
      auto many_i32 = [&]() {
        int acc[16];
        unpack16(pk, acc);
        int done_step = -1;
        int t0 = tok[0], t1 = tok[1], t2 = tok[2];
        for (int k = 0; k < a.nact; ++k) {
          const int cur[3] = {t0, t1, t2};
          if (k + 1 < a.nact) {  // prefetch the next action's tokens
            t0 = tok[3 * (k + 1)];
            t1 = tok[3 * (k + 1) + 1];
            t2 = tok[3 * (k + 1) + 2];
          }
          const S4Factors f = s4_factors<MODE != GENF>(cur, q, a.shift);
          int chg = 0;
          s4_rank1(acc, f, chg);
          if constexpr (MODE == MANY) {
            uint32_t nz = 0;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
              nz |= static_cast<uint32_t>(acc[t]);
              ovf |= acc[t] + 128;
            }
            if (!team_any<4>((nz & 255) != 0) && done_step < 0) done_step = k;
          }
        }
        uint32_t nz = 0;
        const uint4 o = pack16(acc, nz, ovf);
        if (live) *reinterpret_cast<uint4*>(out_blk + out_off) = o;
        const bool any_ovf = team_any<4>((ovf & ~255) != 0);
        if (q == 0 && live) {
          if constexpr (MODE == MANY) (a.done_step + g0)[lg] = done_step;
          if (a.overflow && any_ovf) (a.overflow + g0)[lg] = 1;
        }
      };
      if constexpr (MODE == GENF) {
        many_i32();
      } else {
        // step_many on the saturating int16 lattice (tg_packed.h): x = 256 n + 128 per half, weights
        // 256 w, v_pk_mad_i16 clamp; the zero test is an OR, the int8 range check one test at the end.
        // Operands must be representable (every factor in [-128,127]); teams that are not, or that
        // leave the lattice (an int8 overflow), are redone by many_i32 from the untouched input.
        uint32_t A[8];
        unpack_pairs(pk, A);
#pragma unroll
        for (int p = 0; p < 8; ++p) A[p] = pk_add_u16(pk_lshl8_b16(A[p]), kLatticeZero);
        const uint32_t shp = __builtin_amdgcn_perm(static_cast<uint32_t>(a.shift), static_cast<uint32_t>(a.shift), 0x05040100u);
        uint32_t rng1 = 0, rng2 = 0;  // range accumulators: (x + 128) must stay below 256
        int done_step = -1;
        int t0 = tok[0], t1 = tok[1], t2 = tok[2];
        for (int k = 0; k < a.nact; ++k) {
          const uint32_t du = t0, dv = t1, dw = t2;
          if (k + 1 < a.nact) {
            t0 = tok[3 * (k + 1)];
            t1 = tok[3 * (k + 1) + 1];
            t2 = tok[3 * (k + 1) + 2];
          }
          const int ui = a.shift - __builtin_amdgcn_sbfe(static_cast<int>(du), 8 * q, 8);  // -(u_i)
          rng1 |= static_cast<uint32_t>(ui + 128);
          const uint32_t yv = dv << 8, yw = dw << 8;
          const uint32_t vA = pk_sub_i16(__builtin_amdgcn_perm(dv, yv, 0x0A050804u), shp);  // (v0, v1)
          const uint32_t vB = pk_sub_i16(__builtin_amdgcn_perm(dv, yv, 0x0B070906u), shp);  // (v2, v3)
          uint32_t wA = pk_sub_i16(__builtin_amdgcn_perm(dw, yw, 0x0A050804u), shp);        // (w0, w1)
          uint32_t wB = pk_sub_i16(__builtin_amdgcn_perm(dw, yw, 0x0B070906u), shp);        // (w2, w3)
          rng2 |= pk_add_u16(vA, kLatticeZero) | pk_add_u16(vB, kLatticeZero) | pk_add_u16(wA, kLatticeZero) |
                  pk_add_u16(wB, kLatticeZero);
          const uint32_t uip = __builtin_amdgcn_perm(static_cast<uint32_t>(ui), static_cast<uint32_t>(ui), 0x05040100u);
          const uint32_t uvA = pk_mul_lo_u16(vA, uip), uvB = pk_mul_lo_u16(vB, uip);  // (-u v0, -u v1), (-u v2, -u v3)
          wA = pk_lshl8_b16(wA);
          wB = pk_lshl8_b16(wB);
          const uint32_t pr[4] = {__builtin_amdgcn_perm(uvA, uvA, 0x01000100u), __builtin_amdgcn_perm(uvA, uvA, 0x03020302u),
                                  __builtin_amdgcn_perm(uvB, uvB, 0x01000100u), __builtin_amdgcn_perm(uvB, uvB, 0x03020302u)};
          uint32_t nz = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            A[2 * j] = pk_mad_i16_sat(pr[j], wA, A[2 * j]);
            A[2 * j + 1] = pk_mad_i16_sat(pr[j], wB, A[2 * j + 1]);
            nz |= A[2 * j] | A[2 * j + 1];
          }
          if (!team_any<4>((nz & 0xFF00FF00u) != 0) && done_step < 0) done_step = k;
        }
        uint32_t off = (rng1 & ~0xFFu) | (rng2 & 0xFF00FF00u);
#pragma unroll
        for (int p = 0; p < 8; ++p) off |= (A[p] ^ kLatticeZero) & 0x00FF00FFu;
        const bool bad = team_any<4>(off != 0);
        if (!bad) {
          uint32_t w[4];
#pragma unroll
          for (int d = 0; d < 4; ++d) w[d] = __builtin_amdgcn_perm(A[2 * d + 1], A[2 * d], 0x07050301u);
          if (live) *reinterpret_cast<uint4*>(out_blk + out_off) = uint4{w[0], w[1], w[2], w[3]};
          if (q == 0 && live) (a.done_step + g0)[lg] = done_step;
        } else {
          if (threadIdx.x == (threadIdx.x & ~3)) atomicAdd(&g_fallback_workgroups, 1ull);  // one per team
          many_i32();
        }
      }
    }
  } else {  // EXPAND: child (g, k) lives at out + (g*nact + k) * out_stride
    int8_t* out_blk = a.out + g0 * a.nact * a.out_stride;
    const int64_t c0 = g0 * a.nact;
    // The parent slice stays in registers as 8 int16 pairs; a child costs 8 v_pk_mad_i16 and about 40
    // VALU ops in all.  int16 products need |factor| <= 31; a child with larger factors is redone by
    // its 4-lane team in 32-bit (child_i32).
    uint32_t Pp[8];
    unpack_pairs(pk, Pp);
    const uint32_t shp = __builtin_amdgcn_perm(static_cast<uint32_t>(a.shift), static_cast<uint32_t>(a.shift), 0x05040100u);
    int64_t child_off = static_cast<int64_t>(lg) * a.nact * a.out_stride + 16 * q;  // advanced by out_stride per child
    int t0 = tok[0], t1 = tok[1], t2 = tok[2];
    for (int k = 0; k < a.nact; ++k, child_off += a.out_stride) {
      const uint32_t du = t0, dv = t1, dw = t2;
      if (k + 1 < a.nact) {  // prefetch the next child's tokens
        t0 = tok[3 * (k + 1)];
        t1 = tok[3 * (k + 1) + 1];
        t2 = tok[3 * (k + 1) + 2];
      }
      const uint32_t child = static_cast<uint32_t>(lg) * static_cast<uint32_t>(a.nact) + k;  // < 64 * 4096
      const int ui = a.shift - __builtin_amdgcn_sbfe(static_cast<int>(du), 8 * q, 8);  // -(u_i)
      const uint32_t yv = dv << 8, yw = dw << 8, yu = du << 8;
      const uint32_t vA = pk_sub_i16(__builtin_amdgcn_perm(dv, yv, 0x0A050804u), shp);
      const uint32_t vB = pk_sub_i16(__builtin_amdgcn_perm(dv, yv, 0x0B070906u), shp);
      const uint32_t wA = pk_sub_i16(__builtin_amdgcn_perm(dw, yw, 0x0A050804u), shp);
      const uint32_t wB = pk_sub_i16(__builtin_amdgcn_perm(dw, yw, 0x0B070906u), shp);
      const uint32_t uA = pk_sub_i16(__builtin_amdgcn_perm(du, yu, 0x0A050804u), shp);
      const uint32_t uB = pk_sub_i16(__builtin_amdgcn_perm(du, yu, 0x0B070906u), shp);
      // range: every factor of the child in [-31, 31]  <=>  (f + 31) <= 62 per half; the test is on
      // (f + 32) & ~63 being zero, which admits exactly [-32, 31] (32^3 still fits int16)
      const uint32_t rng = (pk_add_u16(uA, 0x00200020u) | pk_add_u16(uB, 0x00200020u) | pk_add_u16(vA, 0x00200020u) |
                            pk_add_u16(vB, 0x00200020u) | pk_add_u16(wA, 0x00200020u) | pk_add_u16(wB, 0x00200020u)) &
                           0xFFC0FFC0u;
      // null action <=> u, v or w is the zero vector (the whole vector, not this lane's slice)
      const bool nonnull = ((uA | uB) != 0) && ((vA | vB) != 0) && ((wA | wB) != 0);
      uint4 o;
      uint32_t nz = 0, covf = 0;
      if (rng == 0) {  // team-uniform: all four lanes see the same 12 tokens
        const uint32_t uip = __builtin_amdgcn_perm(static_cast<uint32_t>(ui), static_cast<uint32_t>(ui), 0x05040100u);
        const uint32_t uvA = pk_mul_lo_u16(vA, uip), uvB = pk_mul_lo_u16(vB, uip);
        const uint32_t pr[4] = {__builtin_amdgcn_perm(uvA, uvA, 0x01000100u), __builtin_amdgcn_perm(uvA, uvA, 0x03020302u),
                                __builtin_amdgcn_perm(uvB, uvB, 0x01000100u), __builtin_amdgcn_perm(uvB, uvB, 0x03020302u)};
        uint32_t A[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          A[2 * j] = pk_mad_i16(pr[j], wA, Pp[2 * j]);
          A[2 * j + 1] = pk_mad_i16(pr[j], wB, Pp[2 * j + 1]);
        }
        o = pack_pairs(A, nz, covf);
        covf &= 0xFF00FF00u;
      } else {  // exact 32-bit form for this child
        const int cur[3] = {static_cast<int>(du), static_cast<int>(dv), static_cast<int>(dw)};
        const S4Factors f = s4_factors<true>(cur, q, a.shift);
        int acc[16], chg = 0, c32 = 0;
        unpack16(pk, acc);
        s4_rank1(acc, f, chg);
        o = pack16(acc, nz, c32);
        covf = static_cast<uint32_t>(c32) & ~255u;
      }
      if (live) *reinterpret_cast<uint4*>(out_blk + child_off) = o;
      const bool any_nz = team_any<4>(nz != 0);
      const bool any_ovf = team_any<4>(covf != 0);
      if (q == 0 && live) {
        (a.done + c0)[child] = any_nz ? 0 : 1;
        if (a.changed) (a.changed + c0)[child] = nonnull ? 1 : 0;
        if (a.overflow && any_ovf) (a.overflow + c0)[child] = 1;
      }
    }
  }
}

// tg_expand_i8 for S = 4 with one 4-lane team per CHILD: child ch = parent * k + c is just "a step of
// the parent's state with the child's action, written to slot ch", so consecutive teams write
// consecutive 64-byte children and a wavefront's store is 1 KiB of contiguous memory (the
// team-per-parent loop in s4_kernel<EXPAND> writes 64-byte pieces 64 k bytes apart).  The k teams of
// a parent read the same 16-byte parent slices: one request per wavefront, served from L1/L2.  A
// workgroup takes PB = 64 / k whole parents (k <= 64); lc / k by multiplication (recip = ceil(2^16 / k)).
// KEYS (tg_expand_keyed_i8): the 64-bit key of every child leaves with it -- the transposition-table filter of
// extend_tree (act.py:188-195) then needs no second pass over the children.
template <bool NT, bool KEYS = false>
__global__ __launch_bounds__(kBlock) void s4_expand_kernel(ApplyArgs a, int PB, int recip) {
  const int k = a.nact;
  const int lg = threadIdx.x >> 2, q = threadIdx.x & 3;
  const int64_t p0 = static_cast<int64_t>(blockIdx.x) * PB;
  const int nlc = static_cast<int>(min(static_cast<int64_t>(PB), a.B - p0)) * k;  // live children of this workgroup
  const bool live = lg < nlc;
  const int lc = live ? lg : nlc - 1;                      // dead teams shadow the last live child
  const int pl = (lc * recip) >> 16;                       // parent within the workgroup
  const int64_t c0 = p0 * k;                               // first child of the workgroup
  uint32_t du, dv, dw;
  s4_team_tokens(a.actions + c0 * 12, lc, q, du, dv, dw);
  const uint4 par = *reinterpret_cast<const uint4*>(a.in + p0 * a.in_stride +
                                                    (__umul24(pl, static_cast<uint32_t>(a.in_stride)) + 16u * q));
  uint32_t nz = 0;
  int ovf = 0;
  const uint4 o = s4_step_tiered(par, du, dv, dw, q, a.shift, s4_digits_limit(a.shift), nz, ovf);
  int8_t* const dst = a.out + c0 * a.out_stride + (__umul24(lc, static_cast<uint32_t>(a.out_stride)) + 16u * q);
  if (live) {
    if constexpr (NT) store16_nt(dst, o);
    else *reinterpret_cast<uint4*>(dst) = o;
  }
  const bool any_nz = team_any<4>(nz != 0);
  const bool any_ovf = team_any<4>((ovf & ~255) != 0);
  if constexpr (KEYS) {
    // the child's key while its four slices are in registers (tg_hash_u64's definition: slice q = chunk q); the team's
    // sum by two quad-permute exchanges per half
    uint64_t h = hash_chunk(o, q);
    auto quad_xor = [](uint64_t v, auto ctrl) {
      const uint32_t lo = __builtin_amdgcn_mov_dpp(static_cast<uint32_t>(v), decltype(ctrl)::value, 0xf, 0xf, true);
      const uint32_t hi = __builtin_amdgcn_mov_dpp(static_cast<uint32_t>(v >> 32), decltype(ctrl)::value, 0xf, 0xf, true);
      return static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32);
    };
    h += quad_xor(h, std::integral_constant<int, 0xB1>{});  // lanes (1,0,3,2)
    h += quad_xor(h, std::integral_constant<int, 0x4E>{});  // lanes (2,3,0,1)
    if (q == 0 && live) (a.keys + c0)[lc] = hash_finish(h, 64);
  }
  if (q == 0 && live) {
    (a.done + c0)[lc] = any_nz ? 0 : 1;
    if (a.changed) {
      // null action <=> u, v or w is the zero vector <=> all four of its token bytes equal the shift
      const uint32_t zs = (static_cast<uint32_t>(a.shift) & 0xFFu) * 0x01010101u;
      const bool in8 = static_cast<unsigned>(a.shift + 128) < 256u;  // otherwise no token equals the shift
      (a.changed + c0)[lc] = (in8 && (du == zs || dv == zs || dw == zs)) ? 0 : 1;
    }
    if (a.overflow && any_ovf) (a.overflow + c0)[lc] = 1;
  }
}

// =============================================================================================
// tg_step_stream_i8, S = 4: K steps in ONE launch for action blocks that arrive step by step.
// The dependent-launch boundary (1.55 us between two kernels of one stream) is what bounds the
// single-step entry at BASELINE config 2; a stepper that stays resident pays instead its own chain per step:
//   poll ready[k] (sc1 load) -> the 12 token bytes (sc1 loads: the producer is another kernel or the host) ->
//   8 packed MADs per slice -> state + done stored write-through (sc1) -> drain -> progress word (sc1 store).
// Games are independent, so there is NO barrier of any kind: the unit of work and of progress is the WAVEFRONT
// (16 games x NG, four lanes per game, the slices stay in VGPRs for all K steps).  Unit u = global wavefront index
// owns games [u * 16 NG, (u + 1) * 16 NG) and stores k + 1 into progress[u] once step k of its games is visible.
// Every spin is bounded: a wavefront whose ready word never arrives sets *status = 1 and leaves.
// =============================================================================================
struct StreamArgs {
  int8_t* state;
  const int8_t* actions;    // (K, B, 12) step-major
  uint8_t* done;            // (K, B)
  uint8_t* overflow;        // (B), nullable, sticky
  const uint32_t* ready;    // (K), nullable: all blocks valid at launch
  uint32_t* progress;       // (units), nullable
  uint32_t* status;         // (1), nullable
  int64_t B;
  int64_t stride;
  int K;
  int shift;
  uint32_t wait_ticks;  // how long a wavefront waits for a ready word: ticks of s_memrealtime (100 MHz)
};

// The ready-word protocol shared by the four resident steppers (D = steps a wavefront takes at once, <= 8).
// stream_released: how many of ready[kp], ready[kp + 1], ... are set without a gap, given lane's word in v (lanes < D,
// kp + lane < K); wave-uniform, <= D.
template <int D>
__device__ __forceinline__ int stream_released(const StreamArgs& a, uint32_t v, int kp, int lane) {
  const unsigned long long m = __ballot(lane < D && kp + lane < a.K && v != 0);
  return static_cast<int>(__builtin_ctzll(~m));
}
// stream_wait_released: poll until step kp is released (relaxed agent-scope loads: they bypass this CU's L1).  Bounded in
// TIME: the first miss starts a clock on s_memrealtime (100 MHz, one counter for the whole chip), after a.wait_ticks the
// wavefront sets *status and the caller leaves (returns 0).
template <int D>
__device__ __forceinline__ int stream_wait_released(const StreamArgs& a, int kp, int lane) {
  uint64_t t0 = 0;
  for (;;) {
    const uint32_t v = (lane < D && kp + lane < a.K) ? __hip_atomic_load(a.ready + kp + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const int n = stream_released<D>(a, v, kp, lane);
    if (n) return n;
    const uint64_t now = __builtin_amdgcn_s_memrealtime();
    if (t0 == 0) t0 = now;
    if (now - t0 >= a.wait_ticks) {
      if (lane == 0 && a.status) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return 0;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

template <int NG>
__global__ __launch_bounds__(kBlock) void s4_stream_kernel(StreamArgs a) {
  const int lane = threadIdx.x & 63, q = lane & 3, lg = lane >> 2;
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t g0 = unit * (16 * NG);
  if (g0 >= a.B) return;
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.state, 0, static_cast<int>(a.B * a.stride), 0x00027000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(a.done, 0, 0x7fffffff, 0x00027000);
  uint4 pk[NG];
  int64_t g[NG];
  bool live[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    g[n] = g0 + 16 * n + lg;
    live[n] = g[n] < a.B;
    if (!live[n]) g[n] = a.B - 1;  // dead lanes shadow the last game, stores predicated off
    pk[n] = *reinterpret_cast<const uint4*>(a.state + g[n] * a.stride + 16 * q);
  }
  // the slices stay biased for all K steps, each with its L1 norm (s4_step_biased); un-biased when they are stored
  uint32_t l1s[NG];
  int ovfs[NG];  // bits beyond the low byte: an entry left int8 (general form only); written out once per block
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    uint32_t xb[4];
    ovfs[n] = 0;
    l1s[n] = s4_digits_pre(pk[n], xb);
    pk[n] = uint4{xb[0], xb[1], xb[2], xb[3]};
  }
  const int dig_limit = s4_digits_limit(a.shift);
  // The step's chain, in BLOCKS (round 3).  It used to be, per step: poll ready[k] -> the tokens -> arithmetic ->
  // write-through drain -> progress: three memory round trips in a row.  Now a wavefront takes as many steps at once as
  // it has already SEEN released, up to D: the tokens of a block's D steps are requested together, right behind the
  // stores of the previous block (so that block's drain and this block's token round trip overlap), and a poll of the
  // NEXT block's D ready words travels with them.  One round trip per block instead of three per step; progress is
  // published per block.  A producer that releases block k + 1 only after progress[k] (the interactive case) is seen
  // as "one step released": blocks of one, progress per step, the serial order drain -> publish -> spin -> tokens.
  // Invariant: the tokens of step j are requested only after ready[j] was observed set (by an earlier poll).
  // The requests are asm loads with counted waits: vmcnt counts loads and stores together in issue order, so "all but
  // the loads behind them" is exactly the previous block's stores; hipcc's own bookkeeping would drain everything,
  // progress store included, at the loop header.  No asm load is in flight across the loop's back edge.
  // One dword per lane and step (lane q holds dword min(q, 2) of its game's twelve bytes; s4_team_token_bcast).
  static_assert(NG == 1 || NG == 2, "NG = 4 / 8 were retired with the one-game-per-lane kernel");
  constexpr int D = NG == 1 ? 8 : 4;
  uint32_t tk[D][NG], pollv = 0u;
  const uint32_t toff0 = static_cast<uint32_t>(g0 + lg) * 12u + 4u * static_cast<uint32_t>(q < 3 ? q : 2);
  const uint32_t toff_last = static_cast<uint32_t>(a.B - 1) * 12u + 4u * static_cast<uint32_t>(q < 3 ? q : 2);
  // poll of ready[kp + lane], lane < D (with_poll), then the tokens of steps kb .. kb + D - 1 (steps beyond K - 1 repeat
  // the last one; what lies beyond the released steps is loaded and never looked at): sc1 loads, the producer is another agent
  auto request = [&](int kb, int kp, bool with_poll) {
    if (with_poll) {
      const uint32_t* rp = a.ready + kp;
      const uint32_t poff = (lane < D && kp + lane < a.K) ? 4u * lane : 0u;
      asm volatile("global_load_dword %0, %1, %2 sc1" : "=&v"(pollv) : "v"(poff), "s"(rp) : "memory");
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int kd = kb + d < a.K ? kb + d : a.K - 1;
      const int8_t* blk = a.actions + static_cast<int64_t>(kd) * a.B * 12;
#pragma unroll
      for (int n = 0; n < NG; ++n) {
        uint32_t off = toff0 + 192u * n;
        off = off < toff_last ? off : toff_last;  // dead lanes shadow the last game
        asm volatile("global_load_dword %0, %1, %2 sc1" : "=&v"(tk[d][n]) : "v"(off), "s"(blk) : "memory");
      }
    }
  };
  auto arrived = [&]() {  // after the wait that covers them: from here on the registers hold the loaded values
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int n = 0; n < NG; ++n) asm volatile("" : "+v"(tk[d][n]));
    asm volatile("" : "+v"(pollv));
  };
  // how many of ready[kp], ready[kp + 1], ... are set without a gap, given lane's word in v (lanes < D, kp + lane < K)
  auto released = [&](uint32_t v, int kp) { return stream_released<D>(a, v, kp, lane); };
  auto wait_released = [&](int kp) { return stream_wait_released<D>(a, kp, lane); };
  // (the state is in its registers before the first asm load: hipcc waits for its own loads with vmcnt(0) wherever it
  // thinks one may still be pending -- inside the loop that would be every block)
#pragma unroll
  for (int n = 0; n < NG; ++n) asm volatile("" : "+v"(pk[n].x), "+v"(pk[n].y), "+v"(pk[n].z), "+v"(pk[n].w));
  int kb = 0;                                         // first step of the block (uniform)
  int nb = a.ready ? wait_released(0) : (a.K < D ? a.K : D);  // its steps: released, not yet requested
  if (nb == 0) return;
  // STAGGER (round 4).  All resident wavefronts start together and do identical work, so they stay in lockstep: every
  // wavefront of a SIMD waits for its block's tokens at the same time and then all compute at once.  The FIRST block is cut
  // to 1 .. D steps by the workgroup's residency slot on its CU (consecutive workgroups go round the 8 XCDs, then round an
  // XCD's 32 CUs: blockIdx / 256 counts the slots), which spreads the wavefronts of a SIMD over the period: 0.776 -> 0.754 us
  // per step at 131 072 games with ready words, nothing without (same run, A/B).  What bounds this kernel at full occupancy
  // is the number of its small memory operations (the lane kernel below has the ablation), not the phase of its wavefronts.
  if constexpr (D > 1) {
    const int first = 1 + static_cast<int>((blockIdx.x >> 8) & (D - 1));
    nb = nb < first ? nb : first;
  }
  bool fresh = true;  // nothing stored since the last publish (the first block; after the serial order below)
  for (;;) {
    const bool with_poll = a.ready && kb + nb < a.K;  // uniform
    request(kb, kb + nb, with_poll);
    if (a.progress && !fresh) {  // the previous block's stores have left: publish its last step
      if (with_poll) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * NG + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * NG) : "memory");
      if (lane == 0) __hip_atomic_store(a.progress + unit, static_cast<uint32_t>(kb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // the tokens are in; only that progress store may be under way
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    arrived();
    // (round 4) per BLOCK: does any token byte of the team's steps exceed 3?  (lane q holds dword min(q, 2) of every step's
    // twelve bytes; steps beyond nb repeat valid ones, at worst they send a block through the general form needlessly)
    bool wide[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) {
      uint32_t w = 0;
#pragma unroll
      for (int d = 0; d < D; ++d) w |= tk[d][n];
      wide[n] = quad_or(w & 0xFCFCFCFCu) != 0;
    }
    // one step of every game of the wavefront: done[k] from the team's summed L1 norms (two DPP adds); the overflow flags
    // are only ever raised inside the general form and leave once per block
    auto one_step = [&](int d) {
      const int k = kb + d;
#pragma unroll
      for (int n = 0; n < NG; ++n) {
        uint32_t du, dv, dw;
        s4_team_token_bcast(tk[d][n], du, dv, dw);
        s4_step_biased_blk(pk[n], l1s[n], du, dv, dw, q, a.shift, dig_limit, wide[n], ovfs[n]);
        const uint32_t team_l1 = quad_sum(l1s[n]);
        if (live[n] && q == 0)  // write-through (sc1) stores: visible to other agents once this wavefront's vmcnt drains
          __builtin_amdgcn_raw_buffer_store_b8(static_cast<uint8_t>(team_l1 == 0 ? 1 : 0), drs,
                                               static_cast<int>(static_cast<int64_t>(k) * a.B + g[n]), 0, 16);
      }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (d >= nb) break;  // uniform
      one_step(d);
    }
    if (a.overflow) {
#pragma unroll
      for (int n = 0; n < NG; ++n) {
        if (__builtin_expect(quad_or(static_cast<uint32_t>(ovfs[n]) & ~255u) != 0, 0)) {
          if (live[n] && q == 0) a.overflow[g[n]] = 1;
          ovfs[n] = 0;  // (sticky in memory: raised once is enough)
        }
      }
    }
    // the state leaves once per block, as whole 64-byte games (16 games of a wavefront: 1 KiB in a row): nobody may
    // look at it before the block's progress word, and a block of one -- the interactive case -- is the old per-step store
#pragma unroll
    for (int n = 0; n < NG; ++n) {
      typedef unsigned int tg_u32x4 __attribute__((ext_vector_type(4)));
      if (live[n])
        __builtin_amdgcn_raw_buffer_store_b128(tg_u32x4{pk[n].x ^ 0x80808080u, pk[n].y ^ 0x80808080u, pk[n].z ^ 0x80808080u,
                                                        pk[n].w ^ 0x80808080u}, srs,
                                               static_cast<int>(g[n] * a.stride) + 16 * q, 0, 16);
    }
    kb += nb;
    fresh = false;
    if (kb >= a.K) break;
    nb = a.ready ? released(pollv, kb) : (a.K - kb < D ? a.K - kb : D);
    if (nb == 0) {  // nothing released beyond this block yet: the serial order
      if (a.progress) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(a.progress + unit, static_cast<uint32_t>(kb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      fresh = true;
      nb = wait_released(kb);
      if (nb == 0) return;
    }
  }
  if (a.progress) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's last stores have left
    if (lane == 0) __hip_atomic_store(a.progress + unit, static_cast<uint32_t>(a.K), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// =============================================================================================
// tg_step_stream_i8, S = 4, ONE GAME PER LANE (round 4).  Ablation of s4_stream_kernel<1> at BASELINE config 4's share
// (131 072 games resident, tools/stream_ablate.sh): without its arithmetic 0.77 of 0.84 us per step, without the done
// stores 0.57, without any store 0.49 -- the stepper is bound by the NUMBER of small memory operations (a 16-lane byte
// store and a 192-byte token load per 16 games and step, write-through), not by its instructions.  Here a lane owns a
// whole game (sixteen biased dwords X[i][j], the digits are the l index) and a wavefront 64 games:
//   tokens  one global_load_dwordx3 per lane and step: 768 contiguous bytes per wavefront (were 4 x 192);
//   done    one 64-byte row per wavefront and step (were 4 x 16 bytes);
//   state   once per block, transposed through 4 KiB of LDS per wavefront so that every store instruction writes 1 KiB
//           in a row (lane-strided 16-byte pieces would be partial lines);
//   VALU    per step 4 + 4 byte extractions, 4 products G_i = -u_i W, 16 multiply-adds X[i][j] += v_j G_i, 16 v_sad_u8:
//           ~48 instructions for 64 games (the four-lanes-per-game form: ~27 for 16).
// A step some lane's digit form does not cover (tokens > 3, entries near the int8 range, other shifts) is taken by the
// WHOLE wavefront through the general form on an LDS image of its games (s4_step_slice per slice, rolled): exact, rare.
// Protocol (ready / progress / status, blocks of up to D released steps, counted waits) as s4_stream_kernel.
// =============================================================================================
__global__ __launch_bounds__(kBlock) void s4_stream_kernel_lanes(StreamArgs a) {
  typedef unsigned int tg_u32x3 __attribute__((ext_vector_type(3)));
  typedef unsigned int tg_u32x4 __attribute__((ext_vector_type(4)));
  constexpr int D = 8, NW = kBlock / 64;
  constexpr uint32_t BIAS = 0x80808080u;
  constexpr int kDropped = static_cast<int>(0x80000000u);  // a buffer offset beyond every range: the store is dropped
  __shared__ __attribute__((aligned(16))) uint32_t img[NW][64 * 16];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * NW + wave;
  const int64_t g0 = unit * 64;
  if (g0 >= a.B) return;
  // (range-checked buffers: a dead lane's store goes to kDropped instead of being branched around, so the number of
  // memory operations a block issues is exact -- the counted waits below depend on it)
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.state, 0, static_cast<int>(a.B * a.stride), 0x00027000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(a.done, 0, static_cast<int>(static_cast<int64_t>(a.K) * a.B), 0x00027000);
  const bool live = g0 + lane < a.B;
  const int64_t g = live ? g0 + lane : a.B - 1;  // dead lanes shadow the last game
  uint32_t* const row = &img[wave][lane * 16];
  uint32_t x[16];
  {
    const int8_t* src = a.state + g * a.stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + 16 * i);
      x[4 * i] = v.x ^ BIAS, x[4 * i + 1] = v.y ^ BIAS, x[4 * i + 2] = v.z ^ BIAS, x[4 * i + 3] = v.w ^ BIAS;
    }
  }
  auto norm = [&]() {
    uint32_t s0 = 0, s1 = 0;
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      s0 = __builtin_amdgcn_sad_u8(x[e], BIAS, s0);
      s1 = __builtin_amdgcn_sad_u8(x[e + 1], BIAS, s1);
    }
    return s0 + s1;
  };
  uint32_t l1 = norm();
  int ovf = 0;
  const int dig_limit = s4_digits_limit(a.shift);
  const uint32_t shrep = static_cast<uint32_t>(a.shift) * 0x01010101u;
  // TWO sets of token registers: while the block in tkA is worked on, the next block's tokens travel into tkB (a wavefront
  // alone on its SIMD -- 65 536 games -- otherwise waits a memory round trip per block with nothing to issue)
  tg_u32x3 tkA[D], tkB[D];
  uint32_t pollv = 0u;
  const uint32_t toff = static_cast<uint32_t>(g) * 12u;
  // poll of ready[kp + lane], lane < D (with_poll), then the tokens of steps kb .. kb + D - 1 (steps beyond K - 1 repeat the
  // last one; what lies beyond the released steps is loaded and at most OR-ed into the block's `wide` test)
  auto request = [&](tg_u32x3 (&tk)[D], int kb, int kp, bool with_poll) {
    if (with_poll) {
      const uint32_t* rp = a.ready + kp;
      const uint32_t poff = (lane < D && kp + lane < a.K) ? 4u * lane : 0u;
      asm volatile("global_load_dword %0, %1, %2 sc1" : "=&v"(pollv) : "v"(poff), "s"(rp) : "memory");
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int kd = kb + d < a.K ? kb + d : a.K - 1;
      const int8_t* blk = a.actions + static_cast<int64_t>(kd) * a.B * 12;
      asm volatile("global_load_dwordx3 %0, %1, %2 sc1" : "=&v"(tk[d]) : "v"(toff), "s"(blk) : "memory");
    }
  };
  auto arrived = [&](tg_u32x3 (&tk)[D]) {  // behind the wait that covers them: from here on the registers hold the loaded values
#pragma unroll
    for (int d = 0; d < D; ++d) asm volatile("" : "+v"(tk[d]));
    asm volatile("" : "+v"(pollv));
  };
  auto released = [&](uint32_t v, int kp) { return stream_released<D>(a, v, kp, lane); };
  auto wait_released = [&](int kp) { return stream_wait_released<D>(a, kp, lane); };
  auto publish = [&](int k) {
    if (lane == 0) __hip_atomic_store(a.progress + unit, static_cast<uint32_t>(k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // how many steps the block behind step kn - 1 has, from the poll that travelled with the block before it
  auto next_size = [&](int kn) {
    if (kn >= a.K) return 0;
    return a.ready ? released(pollv, kn) : (a.K - kn < D ? a.K - kn : D);
  };
  // the digit form of one step for the lane's game
  auto fast_step = [&](const tg_u32x3& t) {
    const uint32_t W = t.z - shrep;
    uint32_t G[4], vj[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) G[i] = mul_lo_mad(static_cast<uint32_t>(a.shift) - ((t.x >> (8 * i)) & 255u), W);
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(vj[0]) : "v"(t.y), "s"(a.shift));
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(vj[1]) : "v"(t.y), "s"(a.shift));
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(vj[2]) : "v"(t.y), "s"(a.shift));
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(vj[3]) : "v"(t.y), "s"(a.shift));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) x[4 * i + j] += vj[j] * G[i];
    l1 = norm();
  };
  // the general form of one step for every game of the wavefront, on an LDS image (each lane touches its own row only)
  auto general_step = [&](const tg_u32x3& t) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(row + 4 * i) = uint4{x[4 * i] ^ BIAS, x[4 * i + 1] ^ BIAS, x[4 * i + 2] ^ BIAS, x[4 * i + 3] ^ BIAS};
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      uint32_t nz = 0;
      const uint4 r = s4_step_slice(*reinterpret_cast<const uint4*>(row + 4 * i), t.x, t.y, t.z, i, a.shift, nz, ovf);
      *reinterpret_cast<uint4*>(row + 4 * i) = r;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint4 v = *reinterpret_cast<const uint4*>(row + 4 * i);
      x[4 * i] = v.x ^ BIAS, x[4 * i + 1] = v.y ^ BIAS, x[4 * i + 2] = v.z ^ BIAS, x[4 * i + 3] = v.w ^ BIAS;
    }
    l1 = norm();
  };
#pragma unroll
  for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(x[e]));  // the state is in its registers before the first asm load
  int kb = 0;                                                    // first step of the block in tkA (uniform)
  int nb = a.ready ? wait_released(0) : (a.K < D ? a.K : D);     // its steps
  if (nb == 0) return;
  int nb_next = 0;       // steps of the block behind it, as far as they were SEEN released
  int pub = 0;           // > 0: steps [.., pub) are stored but not yet published
  bool have = false;     // tkA holds this block's tokens
  for (;;) {
    if (!have) {  // the serial order (first block; after a spin): request, drain everything, publish what was pending
      request(tkA, kb, kb + nb, a.ready && kb + nb < a.K);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      arrived(tkA);
      if (a.progress && pub) publish(pub);
      pub = 0;
      nb_next = next_size(kb + nb);
    }
    // ---- the next block's tokens set off before this block is worked on (only steps SEEN released are ever requested)
    const int kn = kb + nb;
    const bool pf = nb_next > 0;                              // uniform
    const bool poll2 = a.ready && kn + nb_next < a.K;         // uniform
    if (pf) request(tkB, kn, kn + nb_next, poll2);
    // ---- this block: does any token byte of the lane's steps exceed 3?
    uint32_t wq = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) wq |= tkA[d].x | tkA[d].y | tkA[d].z;
    const bool wide = (wq & 0xFCFCFCFCu) != 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (d >= nb) break;  // uniform
      if (__builtin_expect(__ballot(wide || static_cast<int>(l1) > dig_limit) != 0, 0)) general_step(tkA[d]);
      else fast_step(tkA[d]);
      // write-through (sc1) stores: visible to other agents once this wavefront's vmcnt drains
      __builtin_amdgcn_raw_buffer_store_b8(static_cast<uint8_t>(l1 == 0 ? 1 : 0), drs,
                                           live ? static_cast<int>(static_cast<int64_t>(kb + d) * a.B + g) : kDropped, 0, 16);
    }
    if (a.overflow && __builtin_expect(__ballot((ovf & ~255) != 0) != 0, 0)) {  // (one more store than counted: the waits then
      if (live && (ovf & ~255)) a.overflow[g] = 1;                               // cover one operation more than they need to)
      ovf = 0;  // sticky in memory: raised once is enough
    }
    // the state leaves once per block, transposed through LDS: chunk c = lane + 64 r is 16-byte piece c & 3 of game c >> 2
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(row + 4 * i) = uint4{x[4 * i] ^ BIAS, x[4 * i + 1] ^ BIAS, x[4 * i + 2] ^ BIAS, x[4 * i + 3] ^ BIAS};
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = lane + 64 * r;
      const uint4 v = *reinterpret_cast<const uint4*>(&img[wave][4 * c]);
      const int64_t gg = g0 + (c >> 2);
      __builtin_amdgcn_raw_buffer_store_b128(tg_u32x4{v.x, v.y, v.z, v.w}, srs,
                                             gg < a.B ? static_cast<int>(gg * a.stride) + 16 * (c & 3) : kDropped, 0, 16);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- in flight now, oldest first: [stores of the block before] [tkB's L = D (+1) loads] [this block's nb + 4 stores]
    const bool whole = nb == D;  // (a partial block -- the last one, or a producer releasing step by step -- drains instead)
    if (a.progress && pub) {  // the block before is visible once everything older than tkB's loads has left
      if (!whole || !pf) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (poll2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D + 1 + D + 4) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D + D + 4) : "memory");
      publish(pub);
      pub = 0;
      if (pf) {  // tkB: older than this block's D + 4 stores and that progress store
        if (whole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D + 4 + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else if (pf) {
      if (whole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D + 4) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    pub = kn;
    kb = kn;
    if (kb >= a.K) break;
    if (pf) {
      arrived(tkB);
#pragma unroll
      for (int d = 0; d < D; ++d) tkA[d] = tkB[d];
      nb = nb_next;
      nb_next = next_size(kb + nb);
      have = true;
    } else {  // nothing seen released beyond this block: drain, publish, spin
      if (a.progress) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish(pub);
      }
      pub = 0;
      nb = wait_released(kb);
      if (nb == 0) return;
      have = false;
    }
  }
  if (a.progress) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's last stores have left
    publish(a.K);
  }
}

// =============================================================================================
// S = 16 single step, register only.  One wavefront per game; lane = (r, j) owns the four rows
// (i = r + 4n, j), n = 0..3, i.e. chunks lane + 64 n.  The game's 48 tokens come straight into
// registers (u and w as uniform dwordx4 loads, v_j as a byte), so there is no LDS staging and no
// workgroup barrier: the wavefront's dependency chain is ONE memory round trip, arithmetic, stores.
// (The staged packed_kernel needs three barriers.)  Factors beyond the 16-bit path's range are handled
// by the same wavefront in 32-bit.  Requires 16-byte aligned state and actions.
//
// History: in rounds 2 and 3 the rows the action touches (u_i v_j != 0: 9 % under the reference's factor distribution) were
// COMPACTED into a 64-entry queue of the wavefront in LDS and worked on in one dense pass, because the unpack / multiply-add
// / pack / range test of all four chunks in the packed int16 form was 45 % of the kernel (4.0 us without it at BASELINE
// config 3, 6.6 with).  With the digit form a row costs ~20 instructions and the kernel does every row in its own lane
// again (comment inside); the queue is gone.
// =============================================================================================
// one chunk: x - (-uv) ... i.e. x + uvn * w, uvn = -u_i v_j; saturating int16 form, 32-bit redo when the range test fails
// (wfetch: the game's 16 w tokens again, for the redo only -- keeping them would cost four registers on the common path)
// The digit form of one row (round 3; s4_step_digits has the argument): a row is 16 bytes = four base-256 integers of
// biased digits, the game's w the four integers Wd[d] = w token dword - shift * 0x01010101, and the update of dword d is
// ONE multiply-add, X' = X + uvn * Wd[d] -- exact when no digit leaves [0, 255], which the caller guarantees up front:
// all 48 tokens <= 3 and 0 <= shift <= 3 (wave-uniform, on the scalar unit) bound every |u v w| by F^3, and the row's
// L1 norm (four v_sad_u8) bounds every |x|.  ~20 VALU instructions per row instead of ~32, and the sixteen that build
// the int16 weight pairs leave the kernel's common path altogether.  Returns false when this lane's row is not covered.
__device__ __forceinline__ bool s16_chunk_digits(const uint4& x, int uvn, const uint32_t (&Wd)[4], int limit, uint4& res,
                                                 uint32_t& cnz) {
  const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
  uint32_t o[4], l1 = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t xb = xs[d] ^ 0x80808080u;
    l1 = __builtin_amdgcn_sad_u8(xb, 0x80808080u, l1);
    o[d] = (xb + static_cast<uint32_t>(uvn) * Wd[d]) ^ 0x80808080u;
  }
  res = uint4{o[0], o[1], o[2], o[3]};
  cnz = o[0] | o[1] | o[2] | o[3];
  return static_cast<int>(l1) <= limit;
}

template <class WFetch>
__device__ __forceinline__ uint4 s16_chunk(const uint4& x, int uvn, const uint32_t (&wp)[8], WFetch wfetch, int shift,
                                           bool wide_shift, uint32_t& cnz, uint32_t& ovf) {
  // Saturating int16 form, as in s4_step_slice: with |factor| <= 255 (int8 tokens, |shift| <= 127) the clamped u*v and
  // (u v) w + x are formed exactly or saturate, so everything the 16-bit form cannot represent ends outside int8 --
  // exactly the results that overflow.  No check of the factors; a chunk whose range test fails is redone in 32-bit by
  // its lane (wrapped bytes + flag).
  const int cl = max(-32767, min(32767, uvn));
  const uint32_t pr = __builtin_amdgcn_perm(static_cast<uint32_t>(cl), static_cast<uint32_t>(cl), 0x05040100u);
  uint32_t A[8];
  unpack_pairs(x, A);
#pragma unroll
  for (int p = 0; p < 8; ++p) A[p] = pk_mad_i16_sat(pr, wp[p], A[p]);
  uint32_t c16 = 0;
  cnz = 0;
  uint4 res = pack_pairs(A, cnz, c16);
  if (__builtin_expect(wide_shift || (c16 & 0xFF00FF00u), 0)) {  // rare: exact 32-bit form of this chunk,
    const uint4 wq = wfetch();
    const uint32_t wd[4] = {wq.x, wq.y, wq.z, wq.w};              // one dword at a time (the common path keeps <= 64 VGPRs:
    const uint32_t pd[4] = {x.x, x.y, x.z, x.w};                  // 8 wavefronts per SIMD, cfg3 resident in one round)
    uint32_t rd[4];
    int o32 = 0;
    cnz = 0;
#pragma unroll 1
    for (int d = 0; d < 4; ++d) {
      int e[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        e[t] = sbyte(pd[d], t) + uvn * (sbyte(wd[d], t) - shift);
        o32 |= e[t] + 128;
      }
      rd[d] = pack4(e[0], e[1], e[2], e[3]);
      cnz |= rd[d];
    }
    res = uint4{rd[0], rd[1], rd[2], rd[3]};
    ovf |= static_cast<uint32_t>(o32) & ~255u;
  }
  return res;
}

// LINES: stores at 128-byte-line granularity -- a chunk is stored when any of the eight chunks of its line changed.
// For batches that stream from HBM: a partially written line costs the memory side a read-modify-write (measured at
// 131 072 games: 148 us with 16-byte or 64-byte stores, 130 us with whole lines, although those write 1.7x the bytes).
// Cache-resident batches store only the rows that changed.
// NTL: the state is read by non-temporal loads.  With whole-line stores and a batch beyond the 256 MiB Infinity Cache
// that is worth a quarter of the launch (131 072 games = 512 MiB: 131.5 -> 99.0 us; 262 144 games: 260 -> 232); up to
// ~300 MiB it is neutral to harmful (77 000 games = 301 MiB: 58.7 / 60.3 us, 65 536 games: 50.3 / 52.0, BASELINE config
// 3: 6.0 / 8.5), from 86 000 games = 336 MiB on it wins (80.0 / 66.4): taken from 320 MiB on.  Without whole-line stores
// (the S = 25 step as it was: 16-byte pieces) it gains nothing at any size (143.1 / 143.0 us at 32 768 games).
// DIG: rows go through the digit form first (false only in the A/B library: TG_S16_NO_DIGITS).
template <int MODE, bool LINES, bool NTL = false, bool DIG = true>
__global__ __launch_bounds__(kBlock, LINES ? 6 : 8) void s16_step_kernel(ApplyArgs a) {  // (LINES keeps the inputs to the end)
  static_assert(MODE == STEP, "s16_step_kernel: single step only");
  // (the wavefront index is uniform; saying so lets the game's tokens come by scalar loads)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  int64_t g = static_cast<int64_t>(sweep_index(blockIdx.x, gridDim.x, a.sweep)) * (kBlock / 64) + wave;
  const bool live = g < a.B;
  if (!live) g = a.B - 1;
  const int8_t* tok = a.actions + g * 48;
  const int8_t* src = a.in + g * a.in_stride + 16 * lane;
  // every load of the wavefront is issued before anything is used.  (Four named chunks, not an array: hipcc keeps an
  // array that lives to the end of the LINES variant in scratch.)
  auto ld = [&](const int8_t* q) {
    if constexpr (NTL) {
      const v4u_t v = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(q));
      return uint4{v.x, v.y, v.z, v.w};
    } else {
      return *reinterpret_cast<const uint4*>(q);
    }
  };
  const uint4 p0 = ld(src), p1 = ld(src + 1024), p2 = ld(src + 2048), p3 = ld(src + 3072);
  const uint4 uq = *reinterpret_cast<const uint4*>(tok);
  const uint4 wq = *reinterpret_cast<const uint4*>(tok + 32);
  auto wfetch = [&]() { return *reinterpret_cast<const uint4*>(tok + 32); };
  const int vj = tok[16 + (lane & 15)] - a.shift;
  const int r = lane >> 4;
  uint32_t nz = 0, ovf = 0;
  const bool inplace = a.in == a.out;
  int8_t* const out = a.out + g * a.out_stride;
  const uint32_t shp = (static_cast<uint32_t>(a.shift) & 0xFFFFu) | (static_cast<uint32_t>(a.shift) << 16);
  const bool wide_shift = static_cast<unsigned>(a.shift + 127) > 254u;  // uniform; factors may exceed 255
  // digit form (s16_chunk_digits): its precondition on tokens and shift is wave-uniform -- all 48 tokens come by scalar
  // loads -- so a game either offers it to every row or to none; limit < 0 = not offered
  const uint4 vq = *reinterpret_cast<const uint4*>(tok + 16);
  const uint32_t tok_or = uq.x | uq.y | uq.z | uq.w | vq.x | vq.y | vq.z | vq.w | wq.x | wq.y | wq.z | wq.w;
  const int dig_limit = (DIG && (tok_or & 0xFCFCFCFCu) == 0) ? s4_digits_limit(a.shift) : -1;
  const uint32_t shrep = static_cast<uint32_t>(a.shift) * 0x01010101u;
  const uint32_t Wd[4] = {wq.x - shrep, wq.y - shrep, wq.z - shrep, wq.w - shrep};
  // one row, digit form first; the packed int16 form (its weight pairs built here, off the common path) for the rest
  auto chunk = [&](const uint4& x, int uvn, uint32_t& cnz) {
    uint4 res;
    if (__builtin_expect(s16_chunk_digits(x, uvn, Wd, dig_limit, res, cnz), 1)) return res;
    uint32_t wp[8];
    unpack_pairs(wq, wp);
#pragma unroll
    for (int p = 0; p < 8; ++p) wp[p] = pk_sub_i16(wp[p], shp);
    return s16_chunk(x, uvn, wp, wfetch, a.shift, wide_shift, cnz, ovf);
  };
  auto differs = [](const uint4& x, const uint4& y) { return x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w; };
  // does this lane store chunk (lane, n), given whether it changed and the ballot of the lanes whose chunk n changed?
  auto stores = [&](bool changed, unsigned long long cm) {
    if (!inplace) return true;  // out of place everything is written
    return LINES ? ((cm >> (lane & ~7)) & 0xFFull) != 0 : changed;
  };

  {
    // EVERY row by its own lane (round 3, late).  Rounds 2-3 compacted the rows the action touches (9 %) into a queue of
    // the wavefront in LDS and did the arithmetic in one dense pass: that paid while a row cost ~32 instructions (packed
    // int16 form).  In the digit form a row costs ~20, and the compaction -- ballots, slots, two LDS trips, a divergent
    // dense pass, for the whole-line variants a third trip back to the owners -- costs more than it saves: 5.61 -> 5.51 us
    // at BASELINE config 3, 3.08 -> 2.87 at 2 048 games, equal within 1.5 % from 128 MiB to 2 GiB of states.
    auto one = [&](int n, const uint4& pn, uint32_t udw) {
      const int ui = a.shift - __builtin_amdgcn_sbfe(static_cast<int>(udw), 8 * r, 8);  // -(u_i), i = r + 4 n
      const int uvn = ui * vj;
      uint32_t cnz;
      const uint4 res = chunk(pn, uvn, cnz);
      nz |= cnz;
      // in place, a row the action left as it was needs no store (LINES: unless one of the eight rows of its line changed)
      const bool chg = differs(res, pn);
      const bool st = LINES ? stores(chg, __ballot(chg)) : (!inplace || chg);
      if (live && st) *reinterpret_cast<uint4*>(out + 16 * (lane + 64 * n)) = res;
    };
    one(0, p0, uq.x);
    one(1, p1, uq.y);
    one(2, p2, uq.z);
    one(3, p3, uq.w);
    const bool any_nz0 = __ballot(nz != 0) != 0;
    const bool any_ovf0 = __ballot(ovf != 0) != 0;
    if (lane == 0 && live) {
      a.done[g] = any_nz0 ? 0 : 1;
      if (a.overflow && any_ovf0) a.overflow[g] = 1;
    }
  }
}

// =============================================================================================
// tg_step_emit at S = 16 (round 4): one env step on the history ring AND the (B,T,16,16,16) float model input of the new
// state in one launch, while that output stays in the caches (two launches -- tg_step_i8, then emit_frames_kernel --
// measured 15.9 us at 1 024 games, T = 4, float16, of which the frames alone are 9.0: the step's round trip and a launch
// boundary are what a fused kernel saves; from kStreamOutBytes of output on the frames kernel's write stream is the
// whole cost and the entry stays two launches).
// s16_step_kernel's mapping -- a wavefront per game, lane (r, j) owns rows (i = r + 4 n, j) = chunks lane + 64 n -- so a
// lane's sixteen elements of a chunk leave as 32 (16-bit types) or 64 (float32) contiguous output bytes; frame 0 comes
// from the registers that hold the new head, frame 1 from the registers the step read the old head into, older frames
// from the ring.
// =============================================================================================
template <typename OutT, bool NT>
__device__ __forceinline__ void s16_emit_chunk(OutT* dst, const uint4& q) {  // sixteen int8 -> sixteen OutT at dst
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
  if constexpr (sizeof(OutT) == 4) {
#pragma unroll
    for (int d = 0; d < 4; ++d) s4_emit_f32<NT>(reinterpret_cast<float*>(dst) + 4 * d, w[d]);
  } else {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const uint2 a = s4_cvt16<OutT>(w[2 * hh]), b = s4_cvt16<OutT>(w[2 * hh + 1]);
      const uint4 o{a.x, a.y, b.x, b.y};
      if constexpr (NT) store16_nt(dst + 8 * hh, o);
      else *reinterpret_cast<uint4*>(dst + 8 * hh) = o;
    }
  }
}

template <typename OutT, bool NT>
__global__ __launch_bounds__(kBlock) void s16_step_emit_kernel(StepEmitArgs a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  int64_t g = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave;
  const bool live = g < a.B;
  if (!live) g = a.B - 1;
  const int8_t* tok = a.actions + g * 48;
  int8_t* const game = a.ring + g * a.game_stride;
  const int nxt = a.head + 1 < a.T ? a.head + 1 : 0;
  const int8_t* src = game + a.head * a.frame_stride + 16 * lane;
  const uint4 p0 = *reinterpret_cast<const uint4*>(src), p1 = *reinterpret_cast<const uint4*>(src + 1024),
              p2 = *reinterpret_cast<const uint4*>(src + 2048), p3 = *reinterpret_cast<const uint4*>(src + 3072);
  const uint4 uq = *reinterpret_cast<const uint4*>(tok);
  const uint4 vq = *reinterpret_cast<const uint4*>(tok + 16);
  const uint4 wq = *reinterpret_cast<const uint4*>(tok + 32);
  auto wfetch = [&]() { return *reinterpret_cast<const uint4*>(tok + 32); };
  const int vj = tok[16 + (lane & 15)] - a.shift;
  const int r = lane >> 4;
  uint32_t nz = 0, ovf = 0;
  const uint32_t shp = (static_cast<uint32_t>(a.shift) & 0xFFFFu) | (static_cast<uint32_t>(a.shift) << 16);
  const bool wide_shift = static_cast<unsigned>(a.shift + 127) > 254u;
  const uint32_t tok_or = uq.x | uq.y | uq.z | uq.w | vq.x | vq.y | vq.z | vq.w | wq.x | wq.y | wq.z | wq.w;
  const int dig_limit = (tok_or & 0xFCFCFCFCu) == 0 ? s4_digits_limit(a.shift) : -1;
  const uint32_t shrep = static_cast<uint32_t>(a.shift) * 0x01010101u;
  const uint32_t Wd[4] = {wq.x - shrep, wq.y - shrep, wq.z - shrep, wq.w - shrep};
  auto chunk = [&](const uint4& x, int uvn, uint32_t& cnz) {  // (s16_step_kernel: the digit form first, then the packed int16 form)
    uint4 res;
    if (__builtin_expect(s16_chunk_digits(x, uvn, Wd, dig_limit, res, cnz), 1)) return res;
    uint32_t wp[8];
    unpack_pairs(wq, wp);
#pragma unroll
    for (int p = 0; p < 8; ++p) wp[p] = pk_sub_i16(wp[p], shp);
    return s16_chunk(x, uvn, wp, wfetch, a.shift, wide_shift, cnz, ovf);
  };
  OutT* const out = static_cast<OutT*>(a.out) + g * (static_cast<int64_t>(a.T) * 4096) + 16 * lane;
  int8_t* const dst = game + nxt * a.frame_stride + 16 * lane;
  auto one = [&](int n, const uint4& pn, uint32_t udw) {
    const int ui = a.shift - __builtin_amdgcn_sbfe(static_cast<int>(udw), 8 * r, 8);  // -(u_i), i = r + 4 n
    uint32_t cnz;
    const uint4 res = chunk(pn, ui * vj, cnz);
    nz |= cnz;
    if (live) {
      *reinterpret_cast<uint4*>(dst + 1024 * n) = res;                      // the new head -> ring slot nxt
      s16_emit_chunk<OutT, NT>(out + 1024 * n, res);                        // frame 0
      if (a.T > 1) s16_emit_chunk<OutT, NT>(out + 4096 + 1024 * n, pn);     // frame 1: the old head
    }
  };
  one(0, p0, uq.x);
  one(1, p1, uq.y);
  one(2, p2, uq.z);
  one(3, p3, uq.w);
  if (live) {
    int slot = a.head;
    for (int f = 2; f < a.T; ++f) {  // older frames from the ring
      slot = slot > 0 ? slot - 1 : a.T - 1;
      const int8_t* const old = game + slot * a.frame_stride + 16 * lane;
      const uint4 z0 = *reinterpret_cast<const uint4*>(old), z1 = *reinterpret_cast<const uint4*>(old + 1024),
                  z2 = *reinterpret_cast<const uint4*>(old + 2048), z3 = *reinterpret_cast<const uint4*>(old + 3072);
      OutT* const of = out + static_cast<int64_t>(f) * 4096;
      s16_emit_chunk<OutT, NT>(of, z0);
      s16_emit_chunk<OutT, NT>(of + 1024, z1);
      s16_emit_chunk<OutT, NT>(of + 2048, z2);
      s16_emit_chunk<OutT, NT>(of + 3072, z3);
    }
  }
  const bool any_nz0 = __ballot(nz != 0) != 0;
  const bool any_ovf0 = __ballot(ovf != 0) != 0;
  if (lane == 0 && live) {
    a.done[g] = any_nz0 ? 0 : 1;
    if (a.scalars) a.scalars[g] = a.t_step;
    if (a.overflow && any_ovf0) a.overflow[g] = 1;
  }
}

// =============================================================================================
// tg_step_tracked_i8 at S = 16 (round 3; s25_tracked_kernel in tg_packed.h has the argument): the in-place step that
// loads only the rows the action touches -- row (i, j) changes iff u_i v_j != 0, which the tokens alone decide: ~9 % of the
// rows, in ~28 % of the game's 128-byte lines -- with the number of non-zero entries carried per game.
// One wavefront per game as in s16_step_kernel; the candidate rows' INDICES are compacted into a queue of up to 256
// entries (every row: never flushed), lane k takes entries k, k + 64, ... with all their loads in flight together,
// then per row: count the non-zero bytes, apply (digit form first, packed int16 form behind it), count again, store.
// =============================================================================================
__global__ __launch_bounds__(kBlock, 8) void s16_tracked_kernel(ApplyArgs a, int32_t* nnz) {
  constexpr int NW = kBlock / 64;
  __shared__ __attribute__((aligned(8))) int2 qm[NW][256];  // (row index i * 16 + j, -u_i v_j)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  int64_t g = static_cast<int64_t>(blockIdx.x) * NW + wave;
  const bool live = g < a.B;
  if (!live) g = a.B - 1;
  const int8_t* tok = a.actions + g * 48;
  int8_t* const st = a.out + g * a.out_stride;
  const int nnz_in = nnz[g];
  const uint4 uq = *reinterpret_cast<const uint4*>(tok);
  const uint4 vq = *reinterpret_cast<const uint4*>(tok + 16);
  const uint4 wq = *reinterpret_cast<const uint4*>(tok + 32);
  auto wfetch = [&]() { return *reinterpret_cast<const uint4*>(tok + 32); };
  const int vj = tok[16 + (lane & 15)] - a.shift;
  const int r = lane >> 4;
  const uint32_t shp = (static_cast<uint32_t>(a.shift) & 0xFFFFu) | (static_cast<uint32_t>(a.shift) << 16);
  const bool wide_shift = static_cast<unsigned>(a.shift + 127) > 254u;
  const uint32_t tok_or = uq.x | uq.y | uq.z | uq.w | vq.x | vq.y | vq.z | vq.w | wq.x | wq.y | wq.z | wq.w;
  const int dig_limit = (tok_or & 0xFCFCFCFCu) == 0 ? s4_digits_limit(a.shift) : -1;
  const uint32_t shrep = static_cast<uint32_t>(a.shift) * 0x01010101u;
  const uint32_t Wd[4] = {wq.x - shrep, wq.y - shrep, wq.z - shrep, wq.w - shrep};
  uint32_t ovf = 0;
  auto chunk = [&](const uint4& x, int uvn, uint32_t& cnz) {
    uint4 res;
    if (__builtin_expect(s16_chunk_digits(x, uvn, Wd, dig_limit, res, cnz), 1)) return res;
    uint32_t wp[8];
    unpack_pairs(wq, wp);
#pragma unroll
    for (int p = 0; p < 8; ++p) wp[p] = pk_sub_i16(wp[p], shp);
    return s16_chunk(x, uvn, wp, wfetch, a.shift, wide_shift, cnz, ovf);
  };
  // ---- candidate rows -> the queue (indices only) ----
  const uint32_t ud[4] = {uq.x, uq.y, uq.z, uq.w};
  int total = 0;  // wave-uniform
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int ui = a.shift - __builtin_amdgcn_sbfe(static_cast<int>(ud[n]), 8 * r, 8);  // -(u_i), i = r + 4 n
    const int uvn = ui * vj;
    const bool cand = uvn != 0;
    const unsigned long long m = __ballot(cand);
    const int slot = total + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)));
    if (cand) qm[wave][slot] = int2{lane + 64 * n, uvn};
    total += __builtin_popcountll(m);
  }
  __builtin_amdgcn_wave_barrier();  // (LDS serves one wavefront's accesses in order)
  // ---- dense passes: entries lane, lane + 64, ...; a pass's loads first ----
  int delta = 0;
  const int npass = (total + 63) >> 6;  // uniform; 1 for the reference's factor distribution
  for (int k0 = 0; k0 < npass; k0 += 2) {
    int2 me[2];
    uint4 x[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = lane + 64 * (k0 + k);
      me[k] = qm[wave][e < total ? e : 0];
      if (e >= total) me[k].x = -1;
      x[k] = *reinterpret_cast<const uint4*>(st + 16 * (me[k].x < 0 ? 0 : me[k].x));
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (me[k].x >= 0) {
        uint32_t cnz;
        const uint4 res = chunk(x[k], me[k].y, cnz);
        delta += nz_bytes16(res) - nz_bytes16(x[k]);
        if (live && (res.x != x[k].x || res.y != x[k].y || res.z != x[k].z || res.w != x[k].w))
          *reinterpret_cast<uint4*>(st + 16 * me[k].x) = res;
      }
    }
  }
  delta = wave_sum(delta);
  const bool wovf = __ballot(ovf != 0) != 0;
  if (lane == 0 && live) {
    const int n = nnz_in + delta;
    nnz[g] = n;
    a.done[g] = n == 0 ? 1 : 0;
    if (a.overflow && wovf) a.overflow[g] = 1;
  }
}

// =============================================================================================
// tg_step_stream_i8, S = 16: one wavefront per game, resident for all K steps, the game in REGISTERS -- sixteen VGPRs of biased
// state per lane (lane (r, j) holds rows (i = r + 4 n, j)); 8192 games = 32 wavefronts per CU on 256 CUs: all resident at
// <= 64 VGPRs.  Every step updates all four rows of a lane in the digit form -- no queue, no LDS image, no divergent dense
// pass; the L1 norm of a new row is this step's zero test and the next step's precondition (one bit per row); the game
// is written through once per block.  Rows the digit form does not cover take the packed int16 form inline.
// (Round 2 kept the state in registers too but compacted candidate rows through an LDS queue and OR-ed all sixteen
// registers per step: 3.35 us per step at BASELINE config 3; round 3's first form -- the tracked step on an LDS image of the
// game -- 2.27; this one 1.76: with the biased state a row costs four multiply-adds, four adds and four v_sad_u8, which is
// less than finding out which rows to skip.)
// =============================================================================================
__global__ __launch_bounds__(kBlock, 8) void s16_stream_kernel(StreamArgs a) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr int NW = kBlock / 64;
  constexpr uint32_t BIAS = 0x80808080u;
  constexpr int D = 8;                                         // steps per block (below)
  __shared__ __attribute__((aligned(16))) uint32_t tokbuf[NW][D][12];  // the block's tokens: 48 bytes per step
  const int lane = threadIdx.x & 63;
  // the game index is wave-uniform; say so (readfirstlane): its token and flag addresses stay on the scalar unit
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t g = static_cast<int64_t>(blockIdx.x) * NW + wave;
  if (g >= a.B) return;
  const int r = lane >> 4;
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.state, 0, static_cast<int>(a.B * a.stride), 0x00027000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(a.done, 0, 0x7fffffff, 0x00027000);
  const int soff = static_cast<int>(g * a.stride);
  // ---- the game -> registers, biased (x ^ 0x80808080): lane (r, j) holds rows (i = r + 4 n, j) = chunks lane + 64 n ----
  auto l1_of = [&](const uint4& q) {
    return static_cast<int>(__builtin_amdgcn_sad_u8(q.w, BIAS, __builtin_amdgcn_sad_u8(q.z, BIAS,
                            __builtin_amdgcn_sad_u8(q.y, BIAS, __builtin_amdgcn_sad_u8(q.x, BIAS, 0u)))));
  };
  const int limit = s4_digits_limit(a.shift);
  uint4 x[4];
  uint32_t okbits = 0;  // bit n: row n's L1 norm <= limit (the digit form's precondition for the next step)
  {
    const int8_t* const src = a.state + g * a.stride + 16 * lane;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const uint4 q = *reinterpret_cast<const uint4*>(src + 1024 * n);
      x[n] = uint4{q.x ^ BIAS, q.y ^ BIAS, q.z ^ BIAS, q.w ^ BIAS};
      okbits |= (l1_of(x[n]) <= limit ? 1u : 0u) << n;
    }
  }
  const uint32_t shp = (static_cast<uint32_t>(a.shift) & 0xFFFFu) | (static_cast<uint32_t>(a.shift) << 16);
  const bool wide_shift = static_cast<unsigned>(a.shift + 127) > 254u;  // uniform; factors may exceed 255
  const uint32_t shrep = static_cast<uint32_t>(a.shift) * 0x01010101u;
  // The step's chain runs in BLOCKS as in s4_stream_kernel: up to D steps the wavefront has seen released are taken at
  // once -- their tokens requested together right behind the previous block's stores, with the poll of the next block's
  // ready words; asm loads (sc1: the producer is another agent), counted waits, nothing in flight across the back edge.
  // ONE dword per lane and step -- lane l < 12 asks for dword l of the step's 48 token bytes -- staged through LDS once
  // they are in, so that the steps run as a rolled loop (one copy of the code, no token registers live across it): a
  // step reads u and w back as two uniform 16-byte reads (on to the scalar unit: they are the same for the whole
  // wavefront) and its v_j as a byte.
  uint32_t tk[D], pollv = 0u;
  const uint32_t tk_off = 4u * (lane < 12 ? lane : 11);
  auto tokens_of = [&](int k) { return a.actions + (static_cast<int64_t>(k) * a.B + g) * 48; };
  auto request = [&](int kb, int kp, bool with_poll) {
    if (with_poll) {
      const uint32_t* rp = a.ready + kp;
      const uint32_t poff = (lane < D && kp + lane < a.K) ? 4u * lane : 0u;
      asm volatile("global_load_dword %0, %1, %2 sc1" : "=&v"(pollv) : "v"(poff), "s"(rp) : "memory");
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {  // (steps beyond K - 1 repeat the last one; what lies beyond the released steps is never looked at)
      const int8_t* tp = tokens_of(kb + d < a.K ? kb + d : a.K - 1);
      asm volatile("global_load_dword %0, %1, %2 sc1" : "=&v"(tk[d]) : "v"(tk_off), "s"(tp) : "memory");
    }
  };
  auto arrived = [&]() {
#pragma unroll
    for (int d = 0; d < D; ++d) asm volatile("" : "+v"(tk[d]));
    asm volatile("" : "+v"(pollv));
    if (lane < 12) {
#pragma unroll
      for (int d = 0; d < D; ++d) tokbuf[wave][d][lane] = tk[d];
    }
    __builtin_amdgcn_wave_barrier();  // (LDS serves one wavefront's accesses in order)
  };
  auto released = [&](uint32_t v, int kp) { return stream_released<D>(a, v, kp, lane); };
  auto wait_released = [&](int kp) { return stream_wait_released<D>(a, kp, lane); };
  // (the state is in its registers before the first asm load, or hipcc waits for it -- with vmcnt(0) -- inside the loop)
#pragma unroll
  for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(x[n].x), "+v"(x[n].y), "+v"(x[n].z), "+v"(x[n].w));
  // one step, its tokens in slot d of the block: EVERY row of the lane in the digit form (no compaction: four rows of
  // sixteen bytes, one product -u_i v_j each, the weight integers on the scalar unit)
  auto step = [&](int k, int d) {
    const uint4 u4 = *reinterpret_cast<const uint4*>(&tokbuf[wave][d][0]), w4 = *reinterpret_cast<const uint4*>(&tokbuf[wave][d][8]);
    const uint32_t vdw = tokbuf[wave][d][4 + ((lane & 15) >> 2)];
    const uint32_t us[4] = {static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(u4.x))),
                            static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(u4.y))),
                            static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(u4.z))),
                            static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(u4.w)))};
    const uint32_t ws[4] = {static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w4.x))),
                            static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w4.y))),
                            static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w4.z))),
                            static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w4.w)))};
    const int vj = __builtin_amdgcn_sbfe(static_cast<int>(vdw), 8 * (lane & 3), 8) - a.shift;
    auto wfetch = [&]() { return uint4{ws[0], ws[1], ws[2], ws[3]}; };  // (the 32-bit redo only)
    uint32_t ovf = 0;
    const uint32_t uw_or = us[0] | us[1] | us[2] | us[3] | ws[0] | ws[1] | ws[2] | ws[3];
    const bool small = (uw_or & 0xFCFCFCFCu) == 0 && __ballot((vdw & 0xFCFCFCFCu) != 0) == 0;  // all 48 tokens <= 3 (uniform)
    const uint32_t Wd[4] = {ws[0] - shrep, ws[1] - shrep, ws[2] - shrep, ws[3] - shrep};
    // X + uvn * Wd per dword: v_mad_u64_u32 from the inline constant 0 (full rate; no register pair to set up) and an add
    auto fast_row = [&](const uint4& xb, int uvn) {
      auto dig = [&](uint32_t xd, uint32_t w) {
        uint64_t rr;
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(rr) : "v"(uvn), "s"(w) : "vcc");
        return xd + static_cast<uint32_t>(rr);
      };
      return uint4{dig(xb.x, Wd[0]), dig(xb.y, Wd[1]), dig(xb.z, Wd[2]), dig(xb.w, Wd[3])};
    };
    // a row the digit form does not cover: un-bias, the packed int16 form (32-bit redo behind it), bias again
    auto slow_row = [&](const uint4& xb, int uvn) {
      uint32_t w0 = ws[0], w1 = ws[1], w2 = ws[2], w3 = ws[3];
      asm volatile("" : "+s"(w0), "+s"(w1), "+s"(w2), "+s"(w3));  // (or hipcc builds the weight pairs on the common path)
      uint32_t wp[8], cnz;
      unpack_pairs(uint4{w0, w1, w2, w3}, wp);
#pragma unroll
      for (int p = 0; p < 8; ++p) wp[p] = pk_sub_i16(wp[p], shp);
      const uint4 r4 = s16_chunk(uint4{xb.x ^ BIAS, xb.y ^ BIAS, xb.z ^ BIAS, xb.w ^ BIAS}, uvn, wp, wfetch, a.shift, wide_shift, cnz, ovf);
      return uint4{r4.x ^ BIAS, r4.y ^ BIAS, r4.z ^ BIAS, r4.w ^ BIAS};
    };
    const bool all_fast = small && __ballot((okbits & 15u) != 15u) == 0;  // uniform
    uint32_t l1tot = 0;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int ui = a.shift - __builtin_amdgcn_sbfe(static_cast<int>(us[n]), 8 * r, 8);  // -(u_i), i = r + 4 n
      const int uvn = __mul24(ui, vj);  // (|factor| <= 255 here: full rate, v_mul_lo_u32 is a quarter-rate instruction)
      if (__builtin_expect(all_fast, 1)) x[n] = fast_row(x[n], uvn);
      else if (small && ((okbits >> n) & 1u)) x[n] = fast_row(x[n], uvn);
      else if (uvn != 0) x[n] = slow_row(x[n], uvn);
      const int l1 = l1_of(x[n]);
      l1tot += static_cast<uint32_t>(l1);
      okbits = l1 <= limit ? okbits | (1u << n) : okbits & ~(1u << n);
    }
    const bool any_nz = __ballot(l1tot != 0) != 0;
    if (lane == 0)
      __builtin_amdgcn_raw_buffer_store_b8(static_cast<uint8_t>(any_nz ? 0 : 1), drs,
                                           static_cast<int>(static_cast<int64_t>(k) * a.B + g), 0, 16);
    if (__builtin_expect(ovf != 0, 0) && a.overflow) a.overflow[g] = 1;
  };
  // the game leaves once per block (write-through, sc1), as in s4_stream_kernel
  auto put_state = [&]() {
#pragma unroll
    for (int n = 0; n < 4; ++n)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{x[n].x ^ BIAS, x[n].y ^ BIAS, x[n].z ^ BIAS, x[n].w ^ BIAS}, srs,
                                             soff + 16 * (lane + 64 * n), 0, 16);
  };
  int kb = 0;                                                  // first step of the block (uniform)
  int nb = a.ready ? wait_released(0) : (a.K < D ? a.K : D);   // its steps: released, not yet requested
  if (nb == 0) return;
  bool fresh = true;  // nothing stored since the last publish
  for (;;) {
    const bool with_poll = a.ready && kb + nb < a.K;  // uniform
    request(kb, kb + nb, with_poll);
    if (a.progress && !fresh) {  // the previous block's stores have left: publish its last step
      if (with_poll) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
      if (lane == 0) __hip_atomic_store(a.progress + g, static_cast<uint32_t>(kb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // the tokens are in; only that progress store may be under way
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    arrived();
#pragma unroll 1
    for (int d = 0; d < nb; ++d) step(kb + d, d);
    put_state();
    kb += nb;
    fresh = false;
    if (kb >= a.K) break;
    nb = a.ready ? released(pollv, kb) : (a.K - kb < D ? a.K - kb : D);
    if (nb == 0) {  // nothing released beyond this block yet: the serial order
      if (a.progress) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(a.progress + g, static_cast<uint32_t>(kb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      fresh = true;
      nb = wait_released(kb);
      if (nb == 0) return;
    }
  }
  if (a.progress) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's last stores have left
    if (lane == 0) __hip_atomic_store(a.progress + g, static_cast<uint32_t>(a.K), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// =============================================================================================
// tg_step_stream_i8, S = 25 (round 3): one wavefront per game with the game's 15 625 bytes in REGISTERS for all K steps --
// 80 VGPRs of state per lane, four wavefronts per SIMD: 4 096 games (BASELINE config 5's share of one GPU) are resident at
// once on 256 CUs (LDS could hold 2 560).  Registers cannot be indexed by a lane, so nothing is compacted: every step
// touches all twenty 16-byte chunks of every lane -- in the digit form, with the state kept BIASED (x ^ 0x80808080) between
// steps so that a chunk costs eight multiply-adds, four v_sad_u8 (the new L1 norm: the zero test of this step and the
// precondition of the next, remembered as one bit per chunk) and a compare.
// Layout (the period trick of packed_kernel): lane t < 50 owns chunks t + 50 n, n < 20 (16 * 50 = 800 = 32 rows): its
// 16-byte window starts at byte s = 16 t mod 25 of row r0 = floor(16 t / 25) + 32 n and runs into row r0 + 1 when s > 9 --
// s and the split are lane constants, so the two masked weight integers per dword (W0: the window's bytes in row r0, W1:
// those in row r0 + 1) are built once per step and a chunk needs only its two products -u_i v_j, read from a per-step
// table in LDS at a compile-time offset.  X' = X + uv0 * W0 + uv1 * W1 per dword; exact while no digit leaves [0, 255],
// guaranteed by: all 75 tokens <= 3 and 0 <= shift <= 3 (uniform) and the chunk's L1 norm <= 127 - F^3 (per chunk: the
// bit).  A chunk without its bit is done byte by byte in 32-bit (wrap + overflow flag) by its lane, inline.
// Steps come in blocks, tokens staged through LDS, state written through once per block -- as in s16_stream_kernel.
// =============================================================================================
__global__ __launch_bounds__(kBlock, 4) void s25_stream_kernel(StreamArgs a) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  constexpr int NW = kBlock / 64, D = 8, NSLOT = 20, NCH = 977, TL = 50;
  constexpr uint32_t BIAS = 0x80808080u;
  __shared__ int uvt[NW][656];                                          // -u_i v_j per row 25 i + j; 0 from row 625 on
  __shared__ __attribute__((aligned(4))) uint8_t wext[NW][56];          // the w tokens, periodically extended
  __shared__ __attribute__((aligned(16))) uint32_t tokbuf[NW][D][64];   // the block's tokens: 75 bytes per step (a row per LDS-DMA)
  __shared__ __attribute__((aligned(16))) uint32_t pollbuf[NW][64];     // the next block's ready words
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t g = static_cast<int64_t>(blockIdx.x) * NW + wave;
  if (g >= a.B) return;
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.state, 0, static_cast<int>(a.B * a.stride), 0x00027000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(a.done, 0, 0x7fffffff, 0x00027000);
  const int soff = static_cast<int>(g * a.stride);
  const bool act = lane < TL;
  const uint32_t shrep = static_cast<uint32_t>(a.shift) * 0x01010101u;
  const int limit = s4_digits_limit(a.shift);
  auto l1_of = [&](const uint4& q) {
    return static_cast<int>(__builtin_amdgcn_sad_u8(q.w, BIAS, __builtin_amdgcn_sad_u8(q.z, BIAS,
                            __builtin_amdgcn_sad_u8(q.y, BIAS, __builtin_amdgcn_sad_u8(q.x, BIAS, 0u)))));
  };
  // ---- the game -> registers, biased; the padding behind byte 15 624 (chunk 976 = lane 26, slot 19) is held as zero ----
  uint4 x[NSLOT];
  uint32_t okbits = 0;
  {
    const int8_t* const src = a.state + g * a.stride;
#pragma unroll
    for (int n = 0; n < NSLOT; ++n) {
      const int c = lane + TL * n;
      uint4 q{0, 0, 0, 0};
      if (act && c < NCH) q = *reinterpret_cast<const uint4*>(src + 16 * c);
      if (n == NSLOT - 1 && lane == NCH - 1 - TL * (NSLOT - 1)) {
        q.z &= 0xFFu;
        q.w = 0;
      }
      x[n] = uint4{q.x ^ BIAS, q.y ^ BIAS, q.z ^ BIAS, q.w ^ BIAS};
      okbits |= (l1_of(x[n]) <= limit ? 1u : 0u) << n;
    }
  }
  // Token requests by LDS-DMA (global_load_lds_dword: lane l's dword lands at the row's base + 4 l, no VGPR destination): this
  // kernel runs at its register limit, and a register that an asm load has yet to fill may be copied or spilled by hipcc
  // before the data is there -- LDS cannot.  Counted waits as in the other steppers; M0 (the DMA's LDS base) is saved
  // and restored inside the statement.
  // The DMA moves ALIGNED dwords: a step's 75 token bytes start at any byte address A, so lane l < nd asks for dword l of
  // [A - (A & 3), ...), nd = ceil(((A & 3) + 75) / 4) = 19 or 20, and the step reads its token i at byte (A & 3) + i of the row
  // (actions is 4-byte aligned: nothing in front of the buffer is touched, and behind it at most the rest of the dword that
  // holds the last token -- a fixed 20 dwords would ask for [end, end + 4) of the last game's last step when A & 3 <= 1).
  auto tokens_of = [&](int k) { return a.actions + (static_cast<int64_t>(k) * a.B + g) * 75; };
  auto dma = [&](const void* base, uint32_t voff, const void* lds_row) {
    const uint32_t dst = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_row));
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(dst), "v"(voff), "s"(base) : "memory");
  };
  auto request = [&](int kb, int kp, bool with_poll) {
    if (with_poll) dma(a.ready + kp, (lane < D && kp + lane < a.K) ? 4u * lane : 0u, &pollbuf[wave][0]);
#pragma unroll 1
    for (int d = 0; d < D; ++d) {
      const uintptr_t A = reinterpret_cast<uintptr_t>(tokens_of(kb + d < a.K ? kb + d : a.K - 1));
      const uint32_t nd = (static_cast<uint32_t>(A & 3) + 75u + 3u) >> 2;  // dwords that hold this step's 75 tokens
      dma(reinterpret_cast<const void*>(A & ~static_cast<uintptr_t>(3)), static_cast<uint32_t>(lane) < nd ? 4u * lane : 0u,
          &tokbuf[wave][d][0]);
    }
  };
  auto arrived = [&]() { __builtin_amdgcn_wave_barrier(); };  // behind the counted wait: the rows are in LDS
  auto released = [&](uint32_t v, int kp) { return stream_released<D>(a, v, kp, lane); };
  auto wait_released = [&](int kp) { return stream_wait_released<D>(a, kp, lane); };
  // (the state is in its registers before the first asm load, or hipcc waits for it -- with vmcnt(0) -- inside the loop)
#pragma unroll
  for (int n = 0; n < NSLOT; ++n) asm volatile("" : "+v"(x[n].x), "+v"(x[n].y), "+v"(x[n].z), "+v"(x[n].w));

  // one step, its tokens in slot d of the block
  auto step = [&](int k, int d) {
    // (the lane constants are worked out again in every step, from a lane index hipcc cannot see through: hoisted out of
    // the step loop -- sixteen masks and offsets -- they went to scratch, and every step waited for twenty reloads in a row)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const bool act = ln < TL;
    const int ws = (16 * ln) % 25, r0l = (16 * ln) / 25, k0 = 25 - ws;
    const uint32_t actm = act ? ~0u : 0u;
    const int lane = ln;
    const uint8_t* const tb = reinterpret_cast<const uint8_t*>(&tokbuf[wave][d][0]) + (reinterpret_cast<uintptr_t>(tokens_of(k)) & 3);
    // ---- per-step tables: -u_i v_j for the 625 rows, the extended w, the lane's weight integers ----
    struct __attribute__((packed)) U32 { uint32_t v; };
    uint32_t uw[7], uw_or = 0;  // u's bytes 0..27 on the scalar unit (bytes 25..27 are v tokens)
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      uw[i] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(reinterpret_cast<const U32*>(tb + 4 * i)->v)));
      uw_or |= uw[i];
    }
    const int vtok = static_cast<int8_t>(tb[25 + (lane < 25 ? lane : (lane < TL ? lane - 25 : 0))]);
    const int wtok = static_cast<int8_t>(tb[50 + lane % 25]);
    const int vj = vtok - a.shift;
    // are all 75 tokens <= 3?  (uniform)
    const bool small = (uw_or & 0xFCFCFCFCu) == 0 && __ballot(((vtok | wtok) & ~3) != 0) == 0;
#pragma unroll
    for (int m = 0; m < 13; ++m) {  // rows 2 m (lanes 0..24) and 2 m + 1 (lanes 25..49); "row 25" gives the zeros behind the table
      const int ua = sbyte(uw[(2 * m) >> 2], (2 * m) & 3);
      const int ub = 2 * m + 1 < 25 ? sbyte(uw[(2 * m + 1) >> 2], (2 * m + 1) & 3) : a.shift;
      const int ui = a.shift - (lane < 25 ? ua : ub);
      if (act) uvt[wave][TL * m + lane] = __mul24(ui, vj);
    }
    if (lane < 56) wext[wave][lane] = static_cast<uint8_t>(wtok);
    __builtin_amdgcn_wave_barrier();
    uint32_t W0[4], W1[4];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
      const uint8_t* wp = &wext[wave][ws + 4 * dd];
      const uint32_t wq = static_cast<uint32_t>(wp[0]) | (static_cast<uint32_t>(wp[1]) << 8) | (static_cast<uint32_t>(wp[2]) << 16) |
                          (static_cast<uint32_t>(wp[3]) << 24);
      // byte masks: the window's bytes in row r0 / in row r0 + 1 (both 0 in the idle lanes: their weights are 0)
      const int nbr = k0 - 4 * dd;
      const uint32_t mk0 = (nbr <= 0 ? 0u : (nbr >= 4 ? ~0u : ((1u << (8 * nbr)) - 1u))) & actm, mk1 = ~mk0 & actm;
      W0[dd] = (wq & mk0) - (shrep & mk0);
      W1[dd] = (wq & mk1) - (shrep & mk1);
    }
    bool ovf_any = false;  // uniform (kept off the vector registers: the kernel has none to spare)
    // a chunk byte by byte (its precondition failed, or the step's tokens are not small): exact, wrapped, flagged
    auto slow_chunk = [&](const uint4& xb, int uv0, int uv1) {
      uint32_t q0 = xb.x ^ BIAS, q1 = xb.y ^ BIAS, q2 = xb.z ^ BIAS, q3 = xb.w ^ BIAS, ovf = 0;
      if (!act) uv0 = 0, uv1 = 0;
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {
        uint32_t o = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int b = 4 * it + t;
          const int wv = static_cast<int>(static_cast<int8_t>(wext[wave][ws + b])) - a.shift;
          const int e = sbyte(q0, t) + __mul24(b < k0 ? uv0 : uv1, wv);  // (|u_i v_j| < 2^17, |w_l| <= 255)
          ovf |= static_cast<uint32_t>(e + 128) & ~255u;
          o |= (static_cast<uint32_t>(e) & 255u) << (8 * t);
        }
        q0 = q1, q1 = q2, q2 = q3, q3 = o;  // (rotation: no register is indexed by the loop counter)
      }
      ovf_any |= ovf != 0;
      return uint4{q0 ^ BIAS, q1 ^ BIAS, q2 ^ BIAS, q3 ^ BIAS};
    };
    auto fast_chunk = [&](const uint4& xb, int uv0, int uv1) {
      auto dig = [&](uint32_t xd, uint32_t w0, uint32_t w1) {
        uint64_t r;
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0" : "=&v"(r) : "v"(uv0), "v"(w0), "v"(uv1), "v"(w1) : "vcc");
        return xd + static_cast<uint32_t>(r);
      };
      uint4 o;
      o.x = dig(xb.x, W0[0], W1[0]);
      o.y = dig(xb.y, W0[1], W1[1]);
      o.z = dig(xb.z, W0[2], W1[2]);
      o.w = dig(xb.w, W0[3], W1[3]);
      return o;
    };
    // does every chunk of every lane have its bit (and the step small tokens)?  (uniform)
    const bool all_fast = small && __ballot((okbits & 0xFFFFFu) != 0xFFFFFu) == 0;
    uint32_t l1tot = 0;
    const int* const uvp = &uvt[wave][r0l];
    // (the scheduling fences keep hipcc from hoisting all forty table reads of a step -- and with them forty registers -- to the
    // top: the kernel has 128.  A chunk's bit is read before it is replaced: one register for old and new.)
    if (__builtin_expect(all_fast, 1)) {
      int nx0 = uvp[0], nx1 = uvp[1];  // (a chunk's two products are read one chunk ahead)
#pragma unroll
      for (int n = 0; n < NSLOT; ++n) {
        const int uv0 = nx0, uv1 = nx1;
        if (n + 1 < NSLOT) nx0 = uvp[32 * (n + 1)], nx1 = uvp[32 * (n + 1) + 1];
        x[n] = fast_chunk(x[n], uv0, uv1);
        const int l1 = l1_of(x[n]);
        l1tot += static_cast<uint32_t>(l1);
        okbits = l1 <= limit ? okbits : okbits & ~(1u << n);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int n = 0; n < NSLOT; ++n) {
        const int uv0 = uvp[32 * n], uv1 = uvp[32 * n + 1];
        if (small && ((okbits >> n) & 1u)) x[n] = fast_chunk(x[n], uv0, uv1);
        else x[n] = slow_chunk(x[n], uv0, uv1);
        const int l1 = l1_of(x[n]);
        l1tot += static_cast<uint32_t>(l1);
        okbits = l1 <= limit ? okbits | (1u << n) : okbits & ~(1u << n);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const bool any_nz = __ballot(l1tot != 0) != 0;
    if (lane == 0)
      __builtin_amdgcn_raw_buffer_store_b8(static_cast<uint8_t>(any_nz ? 0 : 1), drs,
                                           static_cast<int>(static_cast<int64_t>(k) * a.B + g), 0, 16);
    if (__ballot(ovf_any) != 0 && a.overflow && lane == 0) a.overflow[g] = 1;
  };
  // the game leaves once per block (write-through, sc1): whole chunks; the last one only up to byte 15 624
  auto put_state = [&]() {
#pragma unroll
    for (int n = 0; n < NSLOT; ++n) {
      const int c = lane + TL * n;
      const u32x4 q{x[n].x ^ BIAS, x[n].y ^ BIAS, x[n].z ^ BIAS, x[n].w ^ BIAS};
      if (n < NSLOT - 1) {
        if (act) __builtin_amdgcn_raw_buffer_store_b128(q, srs, soff + 16 * c, 0, 16);
      } else {
        if (act && c < NCH - 1) __builtin_amdgcn_raw_buffer_store_b128(q, srs, soff + 16 * c, 0, 16);
        if (c == NCH - 1) {
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{q[0], q[1]}, srs, soff + 16 * c, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b8(static_cast<uint8_t>(q[2]), srs, soff + 16 * c + 8, 0, 16);
        }
      }
    }
  };
  int kb = 0;                                                  // first step of the block (uniform)
  int nb = a.ready ? wait_released(0) : (a.K < D ? a.K : D);   // its steps: released, not yet requested
  if (nb == 0) return;
  bool fresh = true;  // nothing stored since the last publish
  for (;;) {
    const bool with_poll = a.ready && kb + nb < a.K;  // uniform
    request(kb, kb + nb, with_poll);
    if (a.progress && !fresh) {  // the previous block's stores have left: publish its last step
      if (with_poll) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
      if (lane == 0) __hip_atomic_store(a.progress + g, static_cast<uint32_t>(kb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // the tokens are in; only that progress store may be under way
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    arrived();
#pragma unroll 1
    for (int d = 0; d < nb; ++d) step(kb + d, d);
    put_state();
    kb += nb;
    fresh = false;
    if (kb >= a.K) break;
    nb = a.ready ? released(pollbuf[wave][lane], kb) : (a.K - kb < D ? a.K - kb : D);
    if (nb == 0) {  // nothing released beyond this block yet: the serial order
      if (a.progress) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(a.progress + g, static_cast<uint32_t>(kb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      fresh = true;
      nb = wait_released(kb);
      if (nb == 0) return;
    }
  }
  if (a.progress) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's last stores have left
    if (lane == 0) __hip_atomic_store(a.progress + g, static_cast<uint32_t>(a.K), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// =============================================================================================
// terminal check / nnz, and reset
// =============================================================================================

// Terminal check + nnz.  A team of `lpg` consecutive lanes (power of two <= 64, chosen on the host
// so that small games do not waste a wavefront: S=4 -> 4 lanes, 16 games per wavefront) owns one
// game; 16-byte loads when the layout allows it (vec16), bytes otherwise.
__global__ __launch_bounds__(kBlock) void done_kernel(const int8_t* state, uint8_t* done, int32_t* nnz,
                                                      int64_t B, int N, int64_t stride, int vec16, int lpg) {
  const int lt = threadIdx.x & (lpg - 1);
  const int64_t team = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / lpg;
  const int64_t nteam = (static_cast<int64_t>(gridDim.x) * kBlock) / lpg;
  const int64_t rounds = (B + nteam - 1) / nteam;  // every lane runs the same number of rounds (shuffles)
  for (int64_t it = 0; it < rounds; ++it) {
    const int64_t g = team + it * nteam;
    const bool live = g < B;
    const int8_t* p = state + (live ? g : B - 1) * stride;
    int cnt = 0, body = 0;
    if (vec16) {
      body = N & ~15;
      const int step = 16 * lpg;
      int e = 16 * lt;
      for (; e + 3 * step < body; e += 4 * step) {  // four chunks in flight per lane (S=16: the game in one round trip)
        const uint4 q0 = *reinterpret_cast<const uint4*>(p + e), q1 = *reinterpret_cast<const uint4*>(p + e + step),
                    q2 = *reinterpret_cast<const uint4*>(p + e + 2 * step), q3 = *reinterpret_cast<const uint4*>(p + e + 3 * step);
        cnt += count_nonzero_bytes(q0) + count_nonzero_bytes(q1) + count_nonzero_bytes(q2) + count_nonzero_bytes(q3);
      }
      for (; e < body; e += step) cnt += count_nonzero_bytes(*reinterpret_cast<const uint4*>(p + e));
    }
    for (int e = body + lt; e < N; e += lpg) cnt += p[e] != 0;
    if (lpg == 64) cnt = wave_sum(cnt);  // (DPP: no LDS round trips; uniform)
    else
      for (int off = lpg >> 1; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lt == 0 && live) {
      done[g] = cnt == 0;
      if (nnz) nnz[g] = cnt;
    }
  }
}

// state_out[b] <- template (S^3 bytes) for b in [first, B).  One thread per 16-byte chunk of the
// whole batch (grid-stride); the template (<= 32 KiB) is served from L1/L2.  vec16 == 0: bytes.
__global__ __launch_bounds__(kBlock) void broadcast_kernel(const int8_t* start, int8_t* out, int64_t first,
                                                           int64_t B, int N, int64_t stride, int vec16) {
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t nthr = static_cast<int64_t>(gridDim.x) * kBlock;
  if (vec16) {
    const int nchunk = (N + 15) >> 4, tail = N & 15;
    const int64_t total = (B - first) * nchunk;
    if (total < (1ll << 31)) {  // the usual case: 32-bit index arithmetic (a 64-bit division is ~100 instructions)
      const uint32_t tot = static_cast<uint32_t>(total), nc = static_cast<uint32_t>(nchunk);
      for (uint32_t idx = static_cast<uint32_t>(tid); idx < tot; idx += static_cast<uint32_t>(nthr)) {
        const uint32_t gi = idx / nc, c = idx - gi * nc;
        int8_t* dst = out + (first + gi) * stride + 16 * c;
        if (tail && c == nc - 1) {
          for (int t = 0; t < tail; ++t) dst[t] = start[16 * c + t];
        } else {
          *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(start + 16 * c);
        }
      }
      return;
    }
    for (int64_t idx = tid; idx < total; idx += nthr) {
      const int64_t g = first + idx / nchunk;
      const int c = static_cast<int>(idx - (g - first) * nchunk);
      int8_t* dst = out + g * stride + 16 * c;
      if (tail && c == nchunk - 1) {
        for (int t = 0; t < tail; ++t) dst[t] = start[16 * c + t];
      } else {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(start + 16 * c);
      }
    }
  } else {
    const int64_t total = (B - first) * N;
    for (int64_t idx = tid; idx < total; idx += nthr) {
      const int64_t g = first + idx / N;
      const int e = static_cast<int>(idx - (g - first) * N);
      out[g * stride + e] = start[e];
    }
  }
}

// <n,n,n> tensor into ONE game slot (reference utils.py:158-160): entry [p][q][r] = 1 iff
// p = a*n+j, q = j*n+c, r = a*n+c for some a,j,c  <=>  p/n == r/n, q%n == r%n, p%n == q/n.
// tg_reset_matmul_i8 writes game 0 with this and broadcasts it to the other games.
__global__ __launch_bounds__(kBlock) void matmul_template_kernel(int8_t* dst, int n) {
  const int S = n * n, N = S * S * S;
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < N; e += gridDim.x * kBlock) {
    const int p = e / (S * S), rem = e - p * S * S, q = rem / S, r = rem - q * S;
    dst[e] = (p / n == r / n) && (q % n == r % n) && (p % n == q / n);
  }
}

// dst[b] <- src[b] for b < B: one thread per 16-byte chunk of a game (the mapping of the step kernels without
// their arithmetic), grid = all chunks.  SH >= 0: chunks per game = 1 << SH (S = 4, 8, 16: shifts instead of a
// division).  The padding between games is neither read nor written.  vec16 == 0: byte granularity.
template <int NT>
__global__ __launch_bounds__(kBlock) void copy_kernel(const int8_t* src, int8_t* dst, int64_t B, int nchunk, int sh,
                                                      int tailb, int64_t sstride, int64_t dstride) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  int64_t g;
  int c;
  if (sh >= 0) {
    g = idx >> sh;
    c = static_cast<int>(idx) & (nchunk - 1);
  } else {
    g = idx / nchunk;
    c = static_cast<int>(idx - g * nchunk);
  }
  if (g >= B) return;
  const int8_t* s = src + g * sstride + 16 * c;
  int8_t* d = dst + g * dstride + 16 * c;
  if (tailb != 0 && c == nchunk - 1) {  // the game's last chunk holds only tailb bytes
    for (int t = 0; t < tailb; ++t) d[t] = s[t];
  } else {
    // (NT as a template parameter: behind a run-time flag hipcc merges the two loads / stores into a plain one)
    uint4 q;
    if constexpr (NT >= 1) {
      const v4u_t v = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(s));
      q = uint4{v.x, v.y, v.z, v.w};
    } else {
      q = *reinterpret_cast<const uint4*>(s);
    }
    if constexpr (NT == 2) store16_nt(d, q);
    else *reinterpret_cast<uint4*>(d) = q;
  }
}

__global__ __launch_bounds__(kBlock) void copy_bytes_kernel(const int8_t* src, int8_t* dst, int64_t B, int N,
                                                            int64_t sstride, int64_t dstride) {
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x)
    for (int e = threadIdx.x; e < N; e += kBlock) dst[b * dstride + e] = src[b * sstride + e];
}

}  // namespace tg

// =============================================================================================
// host side: validation, dispatch, C ABI
// =============================================================================================
static thread_local char g_err[512] = "";

// shared with tg_gen.hip (same library); not part of the C ABI
int tg_internal_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define fail tg_internal_fail

namespace {

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return TG_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }

int validate_common(const char* fn, int64_t B, int S, int64_t stride) {
  if (B < 0) return fail(TG_ERR_INVALID, "%s: B=%lld < 0", fn, (long long)B);
  if (S < 1 || S > TG_MAX_S) return fail(TG_ERR_INVALID, "%s: S=%d outside [1,%d]", fn, S, TG_MAX_S);
  if (stride < (int64_t)S * S * S)
    return fail(TG_ERR_INVALID, "%s: game_stride_bytes=%lld < S^3=%d", fn, (long long)stride, S * S * S);
  return TG_OK;
}

// Per-device caches of device constants.  Entries are written with relaxed atomics: two host threads racing
// on a cold entry both query and store the same value.  Nothing else in the library is mutable host state.
constexpr int kMaxDevices = 64;

int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
  return dev;
}

int device_cu_count() {  // of the current device; 256 on MI355X
  static std::atomic<int> cached[kMaxDevices];
  const int dev = current_device();
  if (dev < 0) return 256;
  int n = cached[dev].load(std::memory_order_relaxed);
  if (!n) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// Workgroups of `kernel` one CU holds at `lds` bytes of dynamic LDS, cached per (kernel instantiation, device):
// the slot packs (lds + 1) << 32 | value, so a different LDS size simply re-queries.  The query is a host-side
// calculation on the code object (no stream operation), so it is legal while `st` is being captured.
struct OccupancySlots {
  std::atomic<uint64_t> v[kMaxDevices];
};
template <typename K>
int resident_per_cu(K kernel, int lds, OccupancySlots& slots) {
  const int dev = current_device();
  const uint64_t tag = (static_cast<uint64_t>(lds) + 1) << 32;
  if (dev >= 0) {
    const uint64_t c = slots.v[dev].load(std::memory_order_relaxed);
    if ((c & ~0xffffffffull) == tag) return static_cast<int>(c & 0xffffffffull);
  }
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, tg::kBlock, lds) != hipSuccess || n < 1) n = 1;
  (void)hipGetLastError();
  if (dev >= 0) slots.v[dev].store(tag | static_cast<uint32_t>(n), std::memory_order_relaxed);
  return n;
}

unsigned capped_grid(int64_t blocks) {
  const int64_t cap = 1 << 20;
  return static_cast<unsigned>(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

template <int MODE>
int launch_apply(const char* fn, const tg::ApplyArgs& a_in, hipStream_t st, bool* keys_fused = nullptr) {
  using namespace tg;
  ApplyArgs a = a_in;
  if (a.B == 0) return TG_OK;
  const bool al = (MODE == GENF || (aligned16(a.in) && a.in_stride % 16 == 0)) && aligned16(a.out) &&
                  a.out_stride % 16 == 0;
  const int64_t B = a.B;
  if constexpr (MODE == EXPAND)
    a.stream_out = (B * a.nact * a.out_stride >= kStreamOutBytes || TG_SWITCH("TG_EXPAND_NT")) && !TG_SWITCH("TG_EXPAND_NO_NT");
  if (al && a.S == 4 && aligned4(a.actions) && a.in_stride < (1 << 20) && a.out_stride < (1 << 20)) {
    const int64_t blocks = (B * 4 + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return fail(TG_ERR_INVALID, "%s: B too large", fn);
    if constexpr (MODE == EXPAND) {
      if (a.nact <= 64 && a.out_stride * 64 < (1 << 24)) {
        const int PB = 64 / a.nact, recip = (65536 + a.nact - 1) / a.nact;
        const int64_t eblocks = (B + PB - 1) / PB;
        if (eblocks > 0x7fffffffLL) return fail(TG_ERR_INVALID, "%s: B too large", fn);
        (void)hipGetLastError();
        if (a.keys) {
          if (a.stream_out) hipLaunchKernelGGL((s4_expand_kernel<true, true>), dim3((unsigned)eblocks), dim3(kBlock), 0, st, a, PB, recip);
          else hipLaunchKernelGGL((s4_expand_kernel<false, true>), dim3((unsigned)eblocks), dim3(kBlock), 0, st, a, PB, recip);
          if (keys_fused) *keys_fused = true;
        } else if (a.stream_out) {
          hipLaunchKernelGGL((s4_expand_kernel<true>), dim3((unsigned)eblocks), dim3(kBlock), 0, st, a, PB, recip);
        } else {
          hipLaunchKernelGGL((s4_expand_kernel<false>), dim3((unsigned)eblocks), dim3(kBlock), 0, st, a, PB, recip);
        }
        return check_launch(fn);
      }
    }
    (void)hipGetLastError();
    if constexpr (MODE == STEP) {
      // Non-temporal state loads from 96 MiB of states on (in place, measured: 64 MiB 23.1 / 24.0 us plain / nt,
      // 128 MiB 48.6 / 44.9, 192 MiB 71.8 / 65.1, 256 MiB 92.4 / 85.4, 512 MiB 205 / 202, 1 GiB 432 / 410, 2 GiB
      // 891 / 820, 4 GiB 1861 / 1666); from kS4TokenWaitBytes on the token is awaited before the slice is requested
      // (nt loads without / with the wait: 256 MiB 85.6 / 92.2 us, 512 MiB 202.0 / 188.3, 1 GiB 402.5 / 394.8, 1.5 GiB
      // 611.3 / 580.2, 2 GiB 814.5 / 793.5).
      const int64_t bytes = B * a.in_stride;
      const bool nt = (bytes >= (96ll << 20) || TG_SWITCH("TG_S4_NT_LOADS"));
      const bool tw = nt && (bytes >= kS4TokenWaitBytes || TG_SWITCH("TG_S4_TOKEN_WAIT"));
      // (a batch that sits in the XCDs' L2s anyway -- BASELINE config 2 -- is swept in one direction)
      const S4StepArgs sa{a.in, a.out, a.actions, a.done, a.overflow, a.B, static_cast<uint32_t>(a.in_stride), a.shift,
                          s4_digits_limit(a.shift), bytes > (16ll << 20) ? a.sweep : 0};
#ifdef TG_AB_SWITCHES
      if (TG_SWITCH("TG_S4_NO_DIGITS")) {  // the packed form alone
        if (tw) hipLaunchKernelGGL((s4_step_kernel<true, true, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, sa);
        else if (nt) hipLaunchKernelGGL((s4_step_kernel<true, false, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, sa);
        else hipLaunchKernelGGL((s4_step_kernel<false, false, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, sa);
        return check_launch(fn);
      }
#endif
      // (fewer resident wavefronts, which helps the S = 16 / 25 steps beyond 1.25 GiB, costs here: 2 GiB of states with
      // 24 / 32 / 48 KB of unused LDS per workgroup: 841 / 942 / 1366 us against 772)
      if (tw) hipLaunchKernelGGL((s4_step_kernel<true, true>), dim3((unsigned)blocks), dim3(kBlock), 0, st, sa);
      else if (nt) hipLaunchKernelGGL((s4_step_kernel<true, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, sa);
      else hipLaunchKernelGGL((s4_step_kernel<false, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, sa);
      return check_launch(fn);
    } else {
      hipLaunchKernelGGL((s4_kernel<MODE>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
      return check_launch(fn);
    }
  }
  // packed int16 path: exact while nact * f^3 <= 32000 for every |factor| <= f (checked on device)
  int flim = 0;
  if constexpr (MODE == MANY) {
    flim = 127;  // lattice form (tg_packed.h): u*v and 256*w must be representable in int16
  } else {
    const int64_t n = (MODE == GENF) ? a.nact : 1;  // STEP, EXPAND: one action per result
    while (flim < 31 && static_cast<int64_t>(flim + 1) * (flim + 1) * (flim + 1) * n <= 32000) ++flim;
  }
#define TG_PACKED(S_, TS_)                                                                      \
  do {                                                                                          \
    const int64_t blocks = (B + PGeo<S_, TS_>::GPB - 1) / PGeo<S_, TS_>::GPB;                   \
    if (blocks > 0x7fffffffLL) return fail(TG_ERR_INVALID, "%s: B too large", fn);              \
    const int at = a.nact < PGeo<S_, TS_>::ATILE ? a.nact : PGeo<S_, TS_>::ATILE;               \
    const int ldsb = packed_lds_bytes<S_, TS_, MODE>(at);                                       \
    (void)hipGetLastError();                                                                    \
    hipLaunchKernelGGL((packed_kernel<S_, TS_, MODE>), dim3((unsigned)blocks), dim3(kBlock),    \
                       ldsb, st, a, flim, at);                                                  \
    return check_launch(fn);                                                                    \
  } while (0)
#define TG_ROWS(S_, TS_)                                                                        \
  do {                                                                                          \
    const int64_t blocks = (B + RGeo<S_, TS_>::GPB - 1) / RGeo<S_, TS_>::GPB;                   \
    if (blocks > 0x7fffffffLL) return fail(TG_ERR_INVALID, "%s: B too large", fn);              \
    const int at = a.nact < RGeo<S_, TS_>::ATILE ? a.nact : RGeo<S_, TS_>::ATILE;               \
    const int ldsb = rows_lds_bytes<S_, TS_, MODE>(at);                                         \
    (void)hipGetLastError();                                                                    \
    hipLaunchKernelGGL((rows_kernel<S_, TS_, MODE>), dim3((unsigned)blocks), dim3(kBlock),      \
                       ldsb, st, a, flim, at);                                                  \
    return check_launch(fn);                                                                    \
  } while (0)
  const bool no_mfma = TG_SWITCH("TG_NO_MFMA");  // A/B switch for measurements
  if constexpr (MODE == GENF) {
    // the accumulation over R is a dense contraction: matrix cores (tg_mfma.h); u*v must fit int8 (checked
    // on device, per game), the transposed factors of one game must fit LDS
    if (aligned16(a.out) && a.out_stride % 16 == 0 && a.nact <= 256 && !no_mfma) {
#define TG_MFMA_K(S_, KS_)                                                                      \
  do {                                                                                           \
    /* a workgroup's set-up (tile offsets, staging addresses) is a third of one game's work: give every */ \
    /* workgroup several games, as many workgroups as the chip holds at once, games split evenly */ \
    static OccupancySlots occ;                                                                   \
    const int per_cu = resident_per_cu(genf_mfma_kernel<S_, KS_>, ldsb, occ);                    \
    const int64_t resident = static_cast<int64_t>(per_cu) * device_cu_count();                   \
    const int64_t per_wg = (B + resident - 1) / resident;                                        \
    const int64_t grid = (B + per_wg - 1) / per_wg;                                              \
    (void)hipGetLastError();                                                                     \
    hipLaunchKernelGGL((genf_mfma_kernel<S_, KS_>), dim3((unsigned)grid), dim3(kBlock), ldsb, st, a, Rp); \
    return check_launch(fn);                                                                     \
  } while (0)
#define TG_MFMA(S_)                                                                              \
  do {                                                                                           \
    const int Rp = (a.nact + 31) & ~31;                                                          \
    const int ldsb = mfma_lds_bytes<S_>(Rp);                                                     \
    if (Rp == 32) TG_MFMA_K(S_, 1);                                                              \
    if (Rp == 64) TG_MFMA_K(S_, 2);                                                              \
    TG_MFMA_K(S_, 0);                                                                            \
  } while (0)
      if (a.S == 9) TG_MFMA(9);
      if (a.S == 16) TG_MFMA(16);
      if (a.S == 25) TG_MFMA(25);
#undef TG_MFMA_K
#undef TG_MFMA
    }
  }
  if constexpr (MODE == MANY) {
    // K fused steps: the final state is one accumulation on the matrix cores; games it cannot certify (a step
    // may have left int8, or the zero state was reached before the last step) are flagged through done_step and
    // redone by the lattice kernels below, launched with only_flagged (tg_mfma.h)
    // Where it pays (measured, tools/sweep_many.py): the per-game set-up (transposed factors, input image,
    // per-action scalars, verdict) outweighs the lattice kernels' K S^3 MACs only for long action lists; beyond
    // K = 127 the overflow bound cannot certify the reference's {-1,0,1} factors any more.
    const bool many_always = TG_SWITCH("TG_MFMA_MANY_ALWAYS");  // tests: every eligible shape
    const bool pays = (a.S == 25 && a.nact >= 3) || (a.S == 16 && a.nact >= 20) || (a.S == 9 && a.nact >= 30);  // up to 256
    // (tools/many_k_sweep.py, profiles/r02_many_k_sweep.txt: the matrix-core pass costs ~44.5 us at S=25 B=4096 and ~41 us at
    //  S=16 B=8192 whatever K is -- staging, the tiles' fixed part, verdict; the lattice kernels 30 / 50 / 52 us at K = 2 / 3 / 4
    //  (S=25), 24 / 35 / 41 / 45 / 55 at K = 8 / 16 / 20 / 24 / 32 (S=16), 58 / 82 / 106 against 88 / 96 / 99 at K = 12 / 24 / 32 (S=9))
    if (al && a.nact <= 256 && !no_mfma && (a.S == 9 || a.S == 16 || a.S == 25) && (pays || many_always)) {
      const int Rp = (a.nact + 31) & ~31;
#define TG_MANY_K(S_, KS_)                                                                       \
  do {                                                                                           \
    const int ldsb = many_mfma_lds_bytes<S_>(Rp);                                                \
    static OccupancySlots occ;                                                                   \
    const int64_t resident = static_cast<int64_t>(resident_per_cu(many_mfma_kernel<S_, KS_>, ldsb, occ)) * device_cu_count(); \
    const int64_t per_wg = (B + resident - 1) / resident;                                        \
    const int64_t grid = (B + per_wg - 1) / per_wg;                                              \
    (void)hipGetLastError();                                                                     \
    hipLaunchKernelGGL((many_mfma_kernel<S_, KS_>), dim3((unsigned)grid), dim3(kBlock), ldsb, st, a, Rp); \
    a.only_flagged = 0;                                                                          \
    if (int rc = check_launch(fn)) return rc;                                                    \
  } while (0)
#define TG_MANY(S_)                                                                              \
  do {                                                                                           \
    if (Rp == 32) TG_MANY_K(S_, 1);                                                              \
    else if (Rp == 64) TG_MANY_K(S_, 2);                                                         \
    else TG_MANY_K(S_, 0);                                                                       \
  } while (0)
      if (a.S == 9) TG_MANY(9);
      if (a.S == 16) TG_MANY(16);
      if (a.S == 25) TG_MANY(25);
#undef TG_MANY
#undef TG_MANY_K
      a.only_flagged = 1;
    }
  }
  const bool no_rows = TG_SWITCH("TG_NO_ROWS");  // A/B switch for measurements
  if constexpr (MODE == MANY || MODE == GENF) {
    // odd S, several actions: each lane owns whole rows (tg_rows.h); the LDS transposition is
    // amortised over the actions
    if (al && flim >= 1 && !no_rows && a.nact >= 3) {
      if (a.S == 9) TG_ROWS(9, 64);
      if (a.S == 25) TG_ROWS(25, 256);
    }
  }
#undef TG_ROWS
  const bool no_s16 = TG_SWITCH("TG_NO_S16_DIRECT");  // A/B switch for measurements
  if constexpr (MODE == STEP) {
    if (al && a.S == 16 && aligned16(a.actions) && !no_s16) {
      const int64_t blocks = (B + 3) / 4;
      if (blocks > 0x7fffffffLL) return fail(TG_ERR_INVALID, "%s: B too large", fn);
      (void)hipGetLastError();
      // whole-line stores pay from ~100 MiB of states on (measured: 6.0 / 7.0 us at 32 MiB, 26.3 / 25.5 at 128 MiB,
      // 50.3 / 47.0 at 256 MiB, 150 / 128 at 512 MiB, 16-byte stores / whole lines)
      // non-temporal state loads where the Infinity Cache can still assist a pass but not hold it: 320 MiB .. 1.25 GiB
      // (round 3 sweep, whole lines without / with them: 512 MiB 129.6 / 99.4 us, 1 GiB 257.6 / 230.0, 1.5 GiB 387.0 / 395.7,
      // 2 GiB 515.5 / 537.3, 4 GiB 1023.5 / 1054.0 -- once the footprint is many times the cache the hint only costs)
      const bool nt_band = B * a.in_stride >= kNtLoadsFromBytes && B * a.in_stride < kNtLoadsToBytes;
      // (as at S = 25: beyond 1.25 GiB five workgroups per CU instead of eight -- 2 GiB of states, dynamic LDS 0 / 8 / 14 / 20 /
      // 26 / 34 KB: 516 / 516 / 515 / 514 / 504 / 506 us)
      int s16_lds_pad = B * a.in_stride >= kNtLoadsToBytes ? 32000 : 0;  // (the kernel has no LDS of its own: 160 KB / 32 000 = 5)
#ifdef TG_AB_SWITCHES
      if (TG_SWITCH("TG_S16_NO_DIGITS")) {  // the packed int16 form alone
        if (nt_band || TG_SWITCH("TG_S16_NT_LOADS"))
          hipLaunchKernelGGL((s16_step_kernel<MODE, true, true, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
        else if (B * a.in_stride >= (96ll << 20) || TG_SWITCH("TG_S16_LINES"))
          hipLaunchKernelGGL((s16_step_kernel<MODE, true, false, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
        else
          hipLaunchKernelGGL((s16_step_kernel<MODE, false, false, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
        return check_launch(fn);
      }
#endif
      if ((nt_band || TG_SWITCH("TG_S16_NT_LOADS")))  // (A/B switches: tests)
        hipLaunchKernelGGL((s16_step_kernel<MODE, true, true>), dim3((unsigned)blocks), dim3(kBlock), s16_lds_pad, st, a);
      else if ((B * a.in_stride >= (96ll << 20) || TG_SWITCH("TG_S16_LINES")))  // (A/B switch: tests at small batches)
        hipLaunchKernelGGL((s16_step_kernel<MODE, true>), dim3((unsigned)blocks), dim3(kBlock), s16_lds_pad, st, a);
      else
        hipLaunchKernelGGL((s16_step_kernel<MODE, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
      return check_launch(fn);
    }
  }
  const bool no_s25 = TG_SWITCH("TG_NO_S25_DIRECT");  // A/B switch for measurements
  const bool no_s9 = TG_SWITCH("TG_NO_S9_DIRECT");    // A/B switch for measurements
  if constexpr (MODE == STEP) {
    if (al && a.S == 9 && a.shift >= -127 && a.shift <= 127 && !no_s9) {
      const int64_t blocks = (B + 15) / 16;  // four wavefronts of four games
      if (blocks > 0x7fffffffLL) return fail(TG_ERR_INVALID, "%s: B too large", fn);
      (void)hipGetLastError();
      hipLaunchKernelGGL(s9_step_kernel<STEP>, dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
      return check_launch(fn);
    }
  }
  if constexpr (MODE == EXPAND) {
    // one 16-lane team per child (s9_step_kernel<EXPAND>): 80 -> 61 us at B = 32 768, k = 8
    if (al && a.S == 9 && a.shift >= -127 && a.shift <= 127 && !no_s9 && B * a.nact < 0x7fffffffLL) {
      const int64_t blocks = (B * a.nact + 15) / 16;
      (void)hipGetLastError();
      hipLaunchKernelGGL(s9_step_kernel<EXPAND>, dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
      return check_launch(fn);
    }
  }
  if constexpr (MODE == STEP) {
    // (|shift| <= 127: factors within +-255, which the 32-bit redo of s25_step_kernel takes from its int16 tables)
    if (al && a.S == 25 && a.shift >= -127 && a.shift <= 127 && B <= 0x7fffffffLL && !no_s25) {
      (void)hipGetLastError();
      // as at S=16: whole-line stores once the batch leaves the caches, non-temporal state loads beyond the Infinity
      // Cache (A/B switches: the variants at test sizes)
      // (round 3 sweep, 16-byte stores / whole lines / whole lines + nt loads: 244 MiB 46.8 / 47.2 / 52.9 us, 488 MiB
      // 135.4 / 129.9 / 99.0, 977 MiB 279.0 / 273.7 / 210.6, 1.46 GiB 428.5 / 439.6 / 464.4, 1.9 GiB 529.7 / 552.2 / 554.7,
      // 3.8 GiB 1089 / 1152 / 1247: beyond 1.25 GiB the plain form is the best one again)
      const int64_t bytes25 = B * a.in_stride;
      // Far beyond the caches FEWER resident workgroups stream better (each workgroup reads one 15.6 KB game: with three per
      // CU instead of seven the HBM side sees fewer concurrent streams): unused dynamic LDS holds the kernel to three.
      // 2 GiB of states, dynamic LDS 0 / 12 / 20 / 24 / 32 / 40 KB (7 / 6 / 5 / 4 / 3 / 3 per CU): 614 / 617 / 597 / 588 / 574 /
      // 572 us; two per CU: 751.  (BASELINE config 5's share, 61 MB: 15.0 / 15.1 / - / 16.1 / 16.1 -- there occupancy wins.)
      int s25_lds_pad = bytes25 >= kNtLoadsToBytes ? 36000 : 0;
      const bool nt_band = bytes25 >= kNtLoadsFromBytes && bytes25 < kNtLoadsToBytes;
      if ((nt_band || TG_SWITCH("TG_S25_NT_LOADS")))
        hipLaunchKernelGGL((s25_step_kernel<true, true>), dim3((unsigned)B), dim3(kBlock), s25_lds_pad, st, a);
      else if (((bytes25 >= (96ll << 20) && bytes25 < kNtLoadsToBytes) || TG_SWITCH("TG_S25_LINES")))
        hipLaunchKernelGGL((s25_step_kernel<true, false>), dim3((unsigned)B), dim3(kBlock), s25_lds_pad, st, a);
      else
        hipLaunchKernelGGL((s25_step_kernel<false, false>), dim3((unsigned)B), dim3(kBlock), s25_lds_pad, st, a);
      return check_launch(fn);
    }
  }
  if (al && flim >= 1) {
    // S=9: a game is only 46 chunks, so a wavefront takes FOUR games (teams of 16 lanes, 9 active, 6
    // chunks per lane): measured 0.48 of the HBM peak at 2^19 games against 0.43 (TS=32) and 0.29 (TS=64)
    if (a.S == 9) TG_PACKED(9, 16);
    if constexpr (MODE == EXPAND) {
      // S = 16: children of 128 MiB and more leave by non-temporal stores; with keys asked for (tg_expand_keyed_i8) they
      // are formed in the same launch while a child is in registers
      const bool keyed = a.keys != nullptr;
      if (a.S == 16 && (a.stream_out || keyed)) {
        const int64_t blocks = (B + PGeo<16, 64>::GPB - 1) / PGeo<16, 64>::GPB;
        if (blocks > 0x7fffffffLL) return fail(TG_ERR_INVALID, "%s: B too large", fn);
        const int at = a.nact < PGeo<16, 64>::ATILE ? a.nact : PGeo<16, 64>::ATILE;
        const int ldsb = packed_lds_bytes<16, 64, MODE>(at);
        (void)hipGetLastError();
        if (keyed) {
          if (a.stream_out) hipLaunchKernelGGL((packed_kernel<16, 64, MODE, true, true>), dim3((unsigned)blocks), dim3(kBlock), ldsb, st, a, flim, at);
          else hipLaunchKernelGGL((packed_kernel<16, 64, MODE, false, true>), dim3((unsigned)blocks), dim3(kBlock), ldsb, st, a, flim, at);
          if (keys_fused) *keys_fused = true;
        } else {
          hipLaunchKernelGGL((packed_kernel<16, 64, MODE, true>), dim3((unsigned)blocks), dim3(kBlock), ldsb, st, a, flim, at);
        }
        return check_launch(fn);
      }
    }
    if constexpr (MODE == EXPAND) {
      if (a.S == 25 && a.keys != nullptr) {  // tg_expand_keyed_i8 at S = 25: the keys from the same launch (a workgroup per parent)
        if (B > 0x7fffffffLL) return fail(TG_ERR_INVALID, "%s: B too large", fn);
        const int at = a.nact < PGeo<25, 256>::ATILE ? a.nact : PGeo<25, 256>::ATILE;
        const int ldsb = packed_lds_bytes<25, 256, MODE, true>(at);
        (void)hipGetLastError();
        hipLaunchKernelGGL((packed_kernel<25, 256, MODE, false, true>), dim3((unsigned)B), dim3(kBlock), ldsb, st, a, flim, at);
        if (keys_fused) *keys_fused = true;
        return check_launch(fn);
      }
    }
    if (a.S == 16) TG_PACKED(16, 64);
    if (a.S == 25) TG_PACKED(25, 256);
  }
#undef TG_PACKED
  (void)hipGetLastError(); hipLaunchKernelGGL((slow_kernel<MODE>), dim3(capped_grid(B)), dim3(kBlock), 0, st, a);
  return check_launch(fn);
}

}  // namespace

// The generator in one kernel (tg_genfused.h); called by tg_gen_demos_i8 (tg_gen.hip).
// Returns 1 = launched, 0 = not applicable (the caller takes the token kernel + tg_gen_from_factors_i8), < 0 = error.
int tg_internal_gen_fused(int8_t* target, int8_t* actions, uint8_t* overflow, const int8_t* basis, int64_t B, int S,
                          int R, const tg::Dist& D, int shift, uint64_t seed, uint64_t gid0, int64_t stride,
                          hipStream_t st) {
  using namespace tg;
  const char* fn = "tg_gen_demos_i8";
  if (TG_SWITCH("TG_NO_FUSED_GEN") || TG_SWITCH("TG_NO_MFMA")) return 0;
  if (!(S == 9 || S == 16 || S == 25) || R > 256 || B == 0) return 0;
  if (!aligned16(target) || stride % 16 != 0) return 0;
  if (!basis) {  // the drawn values are the factors: u * v must fit a byte product, tokens must fit int8
    for (int t = 0; t < D.nv; ++t) {
      const int v = D.val[t];
      if (v > 11 || v < -11 || v + shift > 127 || v + shift < -128) return 0;
    }
  }
  const int Rp = (R + 31) & ~31;
  GenArgs ga{target, actions, overflow, basis, B, stride, seed, gid0, R, shift, D};
  int wgs_override = 0;
#ifdef TG_AB_SWITCHES
  ga.ablate = getenv("TG_GF_ABLATE") ? atoi(getenv("TG_GF_ABLATE")) : 0;
  wgs_override = getenv("TG_GF_WGS") ? atoi(getenv("TG_GF_WGS")) : 0;
#endif
#define TG_GF_K(S_, KS_, BAS_, CHK_)                                                               \
  do {                                                                                             \
    const int ldsb = genfused_lds_bytes<S_>(Rp);                                                   \
    static OccupancySlots occ;                                                                     \
    if (D.nthr == 2 && KS_ != 0) {  /* the reference's three values: the specialised draw evaluation */ \
      static OccupancySlots occ3, occ3l;                                                           \
      constexpr bool kLutShape = !(BAS_) && !(CHK_);  /* values exactly (-1,0,1): byte products by table lookup */ \
      const bool lut = kLutShape && lut_values;                                                    \
      void (*kern)(GenArgs, int) = lut ? gen_fused_kernel<S_, KS_, BAS_, 4, CHK_, true, kLutShape>        \
                                       : gen_fused_kernel<S_, KS_, BAS_, 4, CHK_, true>;           \
      const int per_cu3 = wgs_override > 0 ? wgs_override : resident_per_cu(kern, ldsb, lut ? occ3l : occ3); \
      const int64_t resident3 = static_cast<int64_t>(per_cu3) * device_cu_count() * (wgs_override > 0 ? 1 : 2); \
      const int64_t per_wg3 = (B + resident3 - 1) / resident3;                                     \
      const int64_t grid3 = (B + per_wg3 - 1) / per_wg3;                                           \
      (void)hipGetLastError();                                                                     \
      hipLaunchKernelGGL(kern, dim3((unsigned)grid3), dim3(kBlock), ldsb, st, ga, Rp);             \
      if (int rc = check_launch(fn)) return rc;                                                    \
      return 1;                                                                                    \
    }                                                                                              \
    const int per_cu = wgs_override > 0 ? wgs_override : resident_per_cu(gen_fused_kernel<S_, KS_, BAS_, 4, CHK_>, ldsb, occ); \
    /* twice as many workgroups as fit at once (two games each at B = 4096): the second wave of workgroups fills */ \
    /* the chip as the first ones finish, which evens out the tail (measured: 40 -> 38 us) */     \
    const int64_t resident = static_cast<int64_t>(per_cu) * device_cu_count() * (wgs_override > 0 ? 1 : 2); \
    const int64_t per_wg = (B + resident - 1) / resident;                                          \
    const int64_t grid = (B + per_wg - 1) / per_wg;                                                \
    (void)hipGetLastError();                                                                       \
    hipLaunchKernelGGL((gen_fused_kernel<S_, KS_, BAS_, 4, CHK_>), dim3((unsigned)grid), dim3(kBlock), ldsb, st, ga, Rp); \
    if (int rc = check_launch(fn)) return rc;                                                      \
    return 1;                                                                                      \
  } while (0)
#define TG_GF_B(S_, BAS_, CHK_)                                                                    \
  do {                                                                                             \
    if (Rp == 32) TG_GF_K(S_, 1, BAS_, CHK_);                                                      \
    if (Rp == 64) TG_GF_K(S_, 2, BAS_, CHK_);                                                      \
    TG_GF_K(S_, 0, BAS_, CHK_);                                                                    \
  } while (0)
#define TG_GF(S_)                                                                                  \
  do {                                                                                             \
    if (basis) TG_GF_B(S_, true, true);                                                            \
    if (!in_range) TG_GF_B(S_, false, true);                                                       \
    TG_GF_B(S_, false, false);                                                                     \
  } while (0)
  // without a basis the factors are the drawn values: when R * max|value|^3 <= 127 no entry of a target can leave
  // int8 (the reference's {-1,0,1} up to R = 127) and the tiles need no range tracking
  int fmax = 0;
  for (int t = 0; t < D.nv; ++t) fmax = D.val[t] > fmax ? D.val[t] : (-D.val[t] > fmax ? -D.val[t] : fmax);
  const bool in_range = !basis && static_cast<int64_t>(R) * fmax * fmax * fmax <= 127;
  const bool lut_values = D.nv == 3 && D.val[0] == -1 && D.val[1] == 0 && D.val[2] == 1 && !TG_SWITCH("TG_GF_NO_LUT");
  if (S == 9) TG_GF(9);
  if (S == 16) TG_GF(16);
  TG_GF(25);
#undef TG_GF
#undef TG_GF_B
#undef TG_GF_K
}

extern "C" {

int tg_abi_version(void) { return TG_ABI_VERSION; }

int tg_debug_fallbacks(uint64_t* count) {
  if (!count) return fail(TG_ERR_INVALID, "tg_debug_fallbacks: null pointer");
  unsigned long long v = 0;
  hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(tg::g_fallback_workgroups), sizeof(v));
  if (e != hipSuccess) return fail(TG_ERR_HIP, "tg_debug_fallbacks: %s", hipGetErrorString(e));
  *count = v;
  return TG_OK;
}
int tg_debug_handovers(uint64_t* count) {
  if (!count) return fail(TG_ERR_INVALID, "tg_debug_handovers: null pointer");
  unsigned long long v = 0;
  hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(tg::g_many_handovers), sizeof(v));
  if (e != hipSuccess) return fail(TG_ERR_HIP, "tg_debug_handovers: %s", hipGetErrorString(e));
  *count = v;
  return TG_OK;
}
const char* tg_last_error(void) { return g_err; }

static std::atomic<unsigned> g_sweep{0};  // direction of the next tg_step_i8 sweep (sweep_index)

int tg_step_i8(const int8_t* state_in, int8_t* state_out, const int8_t* actions, uint8_t* done,
               uint8_t* overflow, int64_t B, int S, int64_t game_stride_bytes, int shift,
               tg_stream_t stream) {
  if (int rc = validate_common("tg_step_i8", B, S, game_stride_bytes)) return rc;
  if (B && (!state_in || !state_out || !actions || !done))
    return fail(TG_ERR_INVALID, "tg_step_i8: null pointer");
  tg::ApplyArgs a{state_in, state_out, actions, done, nullptr, nullptr, overflow, B,
                  game_stride_bytes, game_stride_bytes, S, 1, shift};
  a.sweep = static_cast<int>(g_sweep.fetch_add(1u, std::memory_order_relaxed) & 1u);
  return launch_apply<tg::STEP>("tg_step_i8", a, static_cast<hipStream_t>(stream));
}

constexpr uint32_t kStreamWaitTicks = 100000000u;  // 1.0 s of s_memrealtime (100 MHz)

// Units of each streamed-stepper variant this device keeps resident at once (from the occupancy of ITS kernel on THIS device).
// S = 4: NG games x 16 per wavefront (NG = 1, 2: 8 workgroups per CU), or 64 games per wavefront in the
// one-game-per-lane kernel (kStreamLanes; four workgroups per CU at 122 VGPRs).
constexpr int kStreamLanes = 0;
static int64_t stream_units_resident(int S, int ng) {
  static OccupancySlots occ1, occ2, occ16, occ25, occl;
  const int64_t cus = device_cu_count();
  if (S == 16) return cus * 4 * resident_per_cu(tg::s16_stream_kernel, 0, occ16);
  if (S == 25) return cus * 4 * resident_per_cu(tg::s25_stream_kernel, 0, occ25);
  switch (ng) {
    case kStreamLanes: return cus * 4 * resident_per_cu(tg::s4_stream_kernel_lanes, 0, occl);
    case 1: return cus * 4 * resident_per_cu(tg::s4_stream_kernel<1>, 0, occ1);
    default: return cus * 4 * resident_per_cu(tg::s4_stream_kernel<2>, 0, occ2);
  }
}

// S = 4: which kernel takes a resident batch of B games.  From kLanesFrom games on the one-game-per-lane kernel (measured,
// four-lanes-per-game / one-game-per-lane, us per step with ready words and progress: 32 768 games 0.34 / 0.38, 49 152
// 0.38 / 0.39, 65 536 0.42 / 0.40, 98 304 0.58 / 0.47, 131 072 0.82 / 0.48, 262 144 1.97 / 0.90 before its token
// prefetch); below that, and beyond what it holds, the smallest NG whose units all fit.  Returns NG (kStreamLanes for
// the lane kernel) or -1 when no variant keeps B games resident; *most = the largest batch any variant holds.
constexpr int64_t kLanesFrom = 57344;
static int s4_stream_variant(int64_t B, int64_t* units, int* games_per_unit, int64_t* most) {
  int64_t best = 0;
  const int64_t lane_cap = stream_units_resident(4, kStreamLanes);
  best = lane_cap * 64;
  const bool lanes_ok = !TG_SWITCH("TG_STREAM_NO_LANES") && (B + 63) / 64 <= lane_cap;
  if (lanes_ok && (B >= kLanesFrom || TG_SWITCH("TG_STREAM_LANES"))) {
    *units = (B + 63) / 64, *games_per_unit = 64;
    return kStreamLanes;
  }
  for (int ng = 1; ng <= 2; ng *= 2) {  // (NG = 4 / 8 -- 103 / 196 VGPRs, no more resident games than the lane kernel -- went in round 4)
    const int64_t u = (B + 16 * ng - 1) / (16 * ng), cap = stream_units_resident(4, ng);
    if (u <= cap) {
      *units = u, *games_per_unit = 16 * ng;
      return ng;
    }
    best = cap * 16 * ng > best ? cap * 16 * ng : best;
  }
  if (most) *most = best;
  return -1;
}

/* the largest batch tg_step_stream_i8 takes WITH ready words: every unit resident at once on the current device */
int tg_step_stream_capacity(int S, int64_t* games) {
  if (S != 4 && S != 16 && S != 25)
    return fail(TG_ERR_UNSUPPORTED, "tg_step_stream_capacity: S=%d (the streamed stepper is built for S=4, S=16 and S=25)", S);
  if (!games) return fail(TG_ERR_INVALID, "tg_step_stream_capacity: null pointer");
  int64_t most = 0;
  if (S == 4) {
    most = stream_units_resident(4, kStreamLanes) * 64;
    for (int ng = 1; ng <= 2; ng *= 2) {
      const int64_t c = stream_units_resident(4, ng) * 16 * ng;
      most = c > most ? c : most;
    }
  } else {
    most = stream_units_resident(S, 1);  // one wavefront per game
  }
  *games = most;
  return TG_OK;
}

/* units (wavefronts) and games per unit of tg_step_stream_i8 for a batch of B games, or a negative TG_ERR_* */
int tg_step_stream_layout(int64_t B, int S, int64_t* n_units, int* games_per_unit) {
  if (B < 0) return fail(TG_ERR_INVALID, "tg_step_stream_layout: B < 0");
  if (S != 4 && S != 16 && S != 25)
    return fail(TG_ERR_UNSUPPORTED, "tg_step_stream_layout: S=%d (the streamed stepper is built for S=4, S=16 and S=25)", S);
  if (S == 16 || S == 25) {  // one wavefront per game at any B (beyond tg_step_stream_capacity the units run in rounds: no ready words)
    if (n_units) *n_units = B;
    if (games_per_unit) *games_per_unit = 1;
    return TG_OK;
  }
  // S = 4: every wavefront must be resident at once when the producer waits for the whole batch
  int64_t units = 0, most = 0;
  int gpu_ = 0;
  if (s4_stream_variant(B, &units, &gpu_, &most) < 0)
    return fail(TG_ERR_UNSUPPORTED, "tg_step_stream_layout: B=%lld exceeds the %lld games this device keeps resident at once",
                (long long)B, (long long)most);
  if (n_units) *n_units = units;
  if (games_per_unit) *games_per_unit = gpu_;
  return TG_OK;
}

int tg_step_stream_i8(int8_t* state, const int8_t* actions, uint8_t* done, uint8_t* overflow, const uint32_t* ready,
                      uint32_t* progress, uint32_t* status, int64_t B, int S, int K, int64_t game_stride_bytes, int shift,
                      tg_stream_t stream) {
  const char* fn = "tg_step_stream_i8";
  if (int rc = validate_common(fn, B, S, game_stride_bytes)) return rc;
  if (K < 1 || K > (1 << 24)) return fail(TG_ERR_INVALID, "%s: K=%d outside [1,2^24]", fn, K);
  if (B == 0) return TG_OK;
  if (!state || !actions || !done) return fail(TG_ERR_INVALID, "%s: null pointer", fn);
  if (S != 4 && S != 16 && S != 25)
    return fail(TG_ERR_UNSUPPORTED, "%s: S=%d (the streamed stepper is built for S=4, S=16 and S=25)", fn, S);
  int64_t units = B;  // S = 16 / 25: one wavefront per game
  int gpu_ = 1, variant = 1;
  if (S == 4) {
    variant = s4_stream_variant(B, &units, &gpu_, nullptr);
    if (variant < 0) {
      // beyond what the device keeps resident: without ready words no producer can be waiting for the whole batch, so the
      // units (64 games each, the one-game-per-lane kernel) simply run in rounds, every wavefront taking its games
      // through all K steps
      if (ready) return tg_step_stream_layout(B, S, nullptr, nullptr);  // (fails with the message that names the capacity)
      variant = TG_SWITCH("TG_STREAM_NO_LANES") ? 1 : kStreamLanes;
      gpu_ = variant == kStreamLanes ? 64 : 16;
      units = (B + gpu_ - 1) / gpu_;
    }
  }
  // S = 16 / 25 beyond the resident batch run in rounds as well -- and with ready words a producer that releases step k+1
  // only once EVERY unit has published k would never see the later rounds start (each would wait out its whole bound and
  // leave the games at different steps): refused, as tg_step_stream_layout refuses it at S = 4
  if (ready && S != 4 && units > stream_units_resident(S, 1))
    return fail(TG_ERR_UNSUPPORTED, "%s: B=%lld exceeds the %lld games of S=%d this device keeps resident at once; with ready "
                "words every unit must be resident (tg_step_stream_capacity) -- pass ready = NULL to run the batch in rounds",
                fn, (long long)B, (long long)stream_units_resident(S, 1), S);
  // (a single game has no stride to speak of; S = 25 reads the 16-byte chunk that holds the game's last byte: it lies inside
  // the last game's final aligned 16 bytes, and only the game's own 9 bytes of it are ever written)
  if (!aligned16(state) || (game_stride_bytes % 16 != 0 && B > 1) || !(S == 16 ? aligned16(actions) : aligned4(actions)) ||
      B * game_stride_bytes > 0x7fffffffLL || static_cast<int64_t>(K) * B > 0x7fffffffLL ||
      static_cast<unsigned>(shift + 127) > 254u)
    return fail(TG_ERR_UNSUPPORTED, "%s: needs 16-byte aligned states, aligned actions (4 bytes at S=4, 16 at S=16), B*stride and K*B < 2^31, |shift| <= 127", fn);
  if ((ready && (reinterpret_cast<uintptr_t>(ready) & 3)) || (progress && (reinterpret_cast<uintptr_t>(progress) & 3)))
    return fail(TG_ERR_INVALID, "%s: ready / progress must be 4-byte aligned", fn);
  tg::StreamArgs a{state, actions, done, overflow, ready, progress, status, B, game_stride_bytes, K, shift, kStreamWaitTicks};
  const unsigned grid = static_cast<unsigned>((units + 3) / 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  if (S == 16) {
    hipLaunchKernelGGL(tg::s16_stream_kernel, dim3(grid), dim3(tg::kBlock), 0, st, a);
    return check_launch(fn);
  }
  if (S == 25) {
    hipLaunchKernelGGL(tg::s25_stream_kernel, dim3(grid), dim3(tg::kBlock), 0, st, a);
    return check_launch(fn);
  }
  switch (variant) {
    case kStreamLanes: hipLaunchKernelGGL(tg::s4_stream_kernel_lanes, dim3(grid), dim3(tg::kBlock), 0, st, a); break;
    case 1: hipLaunchKernelGGL(tg::s4_stream_kernel<1>, dim3(grid), dim3(tg::kBlock), 0, st, a); break;
    default: hipLaunchKernelGGL(tg::s4_stream_kernel<2>, dim3(grid), dim3(tg::kBlock), 0, st, a); break;
  }
  return check_launch(fn);
}

constexpr int64_t kTrackedSparse25 = 2048;  // games (placed by tools/tracked_time.py sweeps)

int tg_step_tracked_i8(int8_t* state, const int8_t* actions, int32_t* nnz, uint8_t* done, uint8_t* overflow, int64_t B,
                       int S, int64_t game_stride_bytes, int shift, tg_stream_t stream) {
  const char* fn = "tg_step_tracked_i8";
  if (int rc = validate_common(fn, B, S, game_stride_bytes)) return rc;
  if (B == 0) return TG_OK;
  if (!state || !actions || !nnz || !done) return fail(TG_ERR_INVALID, "%s: null pointer", fn);
  if (reinterpret_cast<uintptr_t>(nnz) & 3) return fail(TG_ERR_INVALID, "%s: nnz must be 4-byte aligned", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (S == 25 && aligned16(state) && game_stride_bytes % 16 == 0 && static_cast<unsigned>(shift + 127) <= 254u &&
      B <= 0x7fffffffLL) {
    // sparse kernel from kTrackedSparse25 games on, fewer: the full step's kernel with the count updated (one round trip
    // instead of two).  Measured, tg_step_i8 / full + count / sparse: 512 games 5.0 / 5.8 / 8.1 us, 1 024 6.1 / 7.1 / 8.3,
    // 2 048 8.5 / 10.7 / 10.0, 4 096 14.9 / 17.0 / 13.1, 32 768 102 / - / 65, 139 264 (2 GiB) 574 / - / 303
    tg::ApplyArgs a{state, state, actions, done, nnz, nullptr, overflow, B, game_stride_bytes, game_stride_bytes, S, 1, shift};
    (void)hipGetLastError();
    if ((B >= kTrackedSparse25 || TG_SWITCH("TG_TRACKED_SPARSE")) && !TG_SWITCH("TG_TRACKED_FULL"))
      hipLaunchKernelGGL(tg::s25_tracked_kernel, dim3(static_cast<unsigned>((B + 3) / 4)), dim3(tg::kBlock), 0, st, a, nnz);
    else
      hipLaunchKernelGGL((tg::s25_step_kernel<false, false, true>), dim3(static_cast<unsigned>(B)), dim3(tg::kBlock), 0, st, a);
    return check_launch(fn);
  }
  if (S == 16 && aligned16(state) && aligned16(actions) && game_stride_bytes % 16 == 0 && B <= 0x7fffffffLL) {
    // The sparse kernel pays two dependent round trips for ~28 % of the lines (measured, tg_step_i8 / this: 2 048 games 3.3 /
    // 3.5 us, 8 192 5.8 / 6.5, 12 288 11.1 / 8.4, 32 768 26.7 / 16.6, 131 072 99.7 / 73.7, 2 GiB 504 / 296).  The full step's
    // kernel with the count added was no better at the small end (3.8 / 6.6 us): one kernel for every batch.
    tg::ApplyArgs a{state, state, actions, done, nullptr, nullptr, overflow, B, game_stride_bytes, game_stride_bytes, S, 1, shift};
    (void)hipGetLastError();
    hipLaunchKernelGGL(tg::s16_tracked_kernel, dim3(static_cast<unsigned>((B + 3) / 4)), dim3(tg::kBlock), 0, st, a, nnz);
    return check_launch(fn);
  }
  // other sizes and layouts: the full step, then the count (two launches inside this call; same results)
  if (int rc = tg_step_i8(state, state, actions, done, overflow, B, S, game_stride_bytes, shift, stream)) return rc;
  return tg_done_i8(state, done, nnz, B, S, game_stride_bytes, stream);
}

int tg_step_many_i8(const int8_t* state_in, int8_t* state_out, const int8_t* actions,
                    int32_t* done_step, uint8_t* overflow, int64_t B, int S, int K,
                    int64_t game_stride_bytes, int shift, tg_stream_t stream) {
  if (int rc = validate_common("tg_step_many_i8", B, S, game_stride_bytes)) return rc;
  if (K < 1 || K > TG_MAX_ACTIONS)
    return fail(TG_ERR_INVALID, "tg_step_many_i8: K=%d outside [1,%d]", K, TG_MAX_ACTIONS);
  if (B && (!state_in || !state_out || !actions || !done_step))
    return fail(TG_ERR_INVALID, "tg_step_many_i8: null pointer");
  tg::ApplyArgs a{state_in, state_out, actions, nullptr, done_step, nullptr, overflow, B,
                  game_stride_bytes, game_stride_bytes, S, K, shift};
  return launch_apply<tg::MANY>("tg_step_many_i8", a, static_cast<hipStream_t>(stream));
}

int tg_internal_hash(const int8_t* state, uint64_t* hash_out, int64_t B, int S, int64_t stride, hipStream_t st);  // tg_aux.hip

static int expand_common(const char* fn, const int8_t* state_in, int8_t* state_out, const int8_t* actions, uint8_t* done,
                         uint8_t* changed, uint8_t* overflow, uint64_t* keys_out, int64_t B, int S, int k,
                         int64_t in_stride_bytes, int64_t out_stride_bytes, int shift, tg_stream_t stream) {
  if (int rc = validate_common(fn, B, S, in_stride_bytes)) return rc;
  if (int rc = validate_common(fn, B, S, out_stride_bytes)) return rc;
  if (k < 1 || k > TG_MAX_ACTIONS) return fail(TG_ERR_INVALID, "%s: k=%d outside [1,%d]", fn, k, TG_MAX_ACTIONS);
  if (B && (!state_in || !state_out || !actions || !done)) return fail(TG_ERR_INVALID, "%s: null pointer", fn);
  if (B && state_in == state_out) return fail(TG_ERR_INVALID, "%s: in-place expansion is not defined", fn);
  if (reinterpret_cast<uintptr_t>(keys_out) & 7) return fail(TG_ERR_INVALID, "%s: keys_out must be 8-byte aligned", fn);
  tg::ApplyArgs a{state_in, state_out, actions, done, nullptr, changed, overflow, B,
                  in_stride_bytes, out_stride_bytes, S, k, shift};
  a.keys = keys_out;
  bool fused = false;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int rc = launch_apply<tg::EXPAND>(fn, a, st, &fused)) return rc;
  // kernel families that do not hash the children while they hold them: one pass of the key kernel over the children
  // just written (S = 4, the shape MCTS expansion runs at in the reference, is fused)
  if (keys_out && B && !fused) return tg_internal_hash(state_out, keys_out, B * k, S, out_stride_bytes, st);
  return TG_OK;
}

int tg_expand_i8(const int8_t* state_in, int8_t* state_out, const int8_t* actions, uint8_t* done,
                 uint8_t* changed, uint8_t* overflow, int64_t B, int S, int k,
                 int64_t in_stride_bytes, int64_t out_stride_bytes, int shift, tg_stream_t stream) {
  return expand_common("tg_expand_i8", state_in, state_out, actions, done, changed, overflow, nullptr, B, S, k,
                       in_stride_bytes, out_stride_bytes, shift, stream);
}

int tg_expand_keyed_i8(const int8_t* state_in, int8_t* state_out, const int8_t* actions, uint8_t* done,
                       uint8_t* changed, uint8_t* overflow, uint64_t* keys_out, int64_t B, int S, int k,
                       int64_t in_stride_bytes, int64_t out_stride_bytes, int shift, tg_stream_t stream) {
  return expand_common("tg_expand_keyed_i8", state_in, state_out, actions, done, changed, overflow, keys_out, B, S, k,
                       in_stride_bytes, out_stride_bytes, shift, stream);
}

int tg_emit_frames(const int8_t* ring, void* out, float* scalars, int out_dtype, int64_t B, int S, int T, int head_slot,
                   float t_step, int64_t frame_stride_bytes, int64_t game_stride_bytes, tg_stream_t stream);  // tg_aux.hip

int tg_step_emit(int8_t* ring, const int8_t* actions, void* out, float* scalars, uint8_t* done, uint8_t* overflow,
                 int out_dtype, int64_t B, int S, int T, int head_slot, float t_step, int64_t frame_stride_bytes,
                 int64_t game_stride_bytes, int shift, tg_stream_t stream) {
  const char* fn = "tg_step_emit";
  if (int rc = validate_common(fn, B, S, frame_stride_bytes)) return rc;
  if (T < 1 || T > 64 || head_slot < 0 || head_slot >= T)
    return fail(TG_ERR_INVALID, "%s: need 1 <= T <= 64 and 0 <= head_slot < T", fn);
  if (game_stride_bytes < static_cast<int64_t>(T - 1) * frame_stride_bytes + static_cast<int64_t>(S) * S * S)
    return fail(TG_ERR_INVALID, "%s: game_stride_bytes too small for T frames", fn);
  if (out_dtype < 0 || out_dtype > 2) return fail(TG_ERR_INVALID, "%s: out_dtype must be 0 (f32), 1 (f16) or 2 (bf16)", fn);
  if (B == 0) return TG_OK;
  if (!ring || !actions || !out || !done) return fail(TG_ERR_INVALID, "%s: null pointer", fn);
  if (reinterpret_cast<uintptr_t>(out) & 15) return fail(TG_ERR_INVALID, "%s: out must be 16-byte aligned", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nxt = head_slot + 1 < T ? head_slot + 1 : 0;
  // One launch while the model input stays in the caches (65 536 games, T = 4: 9.9 us against 14.3 for step + frames in
  // float16, 14.4 against 18.4 in float32).  Once the output streams to HBM the frames kernel's fully coalesced
  // 16-byte-per-thread write stream wins over the fused kernel's 64-byte team runs (2^20 games: 243 against 185 us,
  // 350 against 320), so from kStreamOutBytes of output on this entry is the two launches.
  const int64_t out_bytes = B * T * 64 * (out_dtype ? 2 : 4);
  const bool fused = S == 4 && (reinterpret_cast<uintptr_t>(ring) & 3) == 0 && frame_stride_bytes % 4 == 0 &&
                     game_stride_bytes % 4 == 0 && aligned4(actions) && static_cast<unsigned>(shift + 127) <= 254u &&
                     (out_bytes < tg::kStreamOutBytes || TG_SWITCH("TG_STEP_EMIT_FUSED"));
  if (fused) {
    tg::StepEmitArgs a{ring, actions, out, scalars, done, overflow, B, frame_stride_bytes, game_stride_bytes, T, head_slot, shift, t_step};
    const int64_t blocks = (B * 4 + tg::kBlock - 1) / tg::kBlock;
    if (blocks > 0x7fffffffLL) return fail(TG_ERR_INVALID, "%s: B too large", fn);
    const bool nt = out_bytes >= tg::kStreamOutBytes || TG_SWITCH("TG_EMIT_NT");  // (A/B library only, see above)
    const dim3 grid(static_cast<unsigned>(blocks)), block(tg::kBlock);
    (void)hipGetLastError();
#define TG_SE(OutT_)                                                                              \
  do {                                                                                            \
    if (nt) hipLaunchKernelGGL((tg::s4_step_emit_kernel<OutT_, true>), grid, block, 0, st, a);   \
    else hipLaunchKernelGGL((tg::s4_step_emit_kernel<OutT_, false>), grid, block, 0, st, a);     \
  } while (0)
    if (out_dtype == 1) TG_SE(__half);
    else if (out_dtype == 2) TG_SE(__hip_bfloat16);
    else TG_SE(float);
#undef TG_SE
    return check_launch(fn);
  }
  // S = 16 while the output stays in the caches: one launch (s16_step_emit_kernel)
  const int64_t out_bytes16 = B * T * 4096 * (out_dtype ? 2 : 4);
  const bool fused16 = S == 16 && aligned16(ring) && frame_stride_bytes % 16 == 0 && game_stride_bytes % 16 == 0 && aligned16(actions) &&
                       (out_bytes16 < tg::kStreamOutBytes || TG_SWITCH("TG_STEP_EMIT_FUSED"));
  if (fused16) {
    tg::StepEmitArgs a{ring, actions, out, scalars, done, overflow, B, frame_stride_bytes, game_stride_bytes, T, head_slot, shift, t_step};
    const int64_t blocks = (B + 3) / 4;
    if (blocks > 0x7fffffffLL) return fail(TG_ERR_INVALID, "%s: B too large", fn);
    const bool nt = out_bytes16 >= tg::kStreamOutBytes;
    const dim3 grid(static_cast<unsigned>(blocks)), block(tg::kBlock);
    (void)hipGetLastError();
#define TG_SE16(OutT_)                                                                             \
  do {                                                                                             \
    if (nt) hipLaunchKernelGGL((tg::s16_step_emit_kernel<OutT_, true>), grid, block, 0, st, a);   \
    else hipLaunchKernelGGL((tg::s16_step_emit_kernel<OutT_, false>), grid, block, 0, st, a);     \
  } while (0)
    if (out_dtype == 1) TG_SE16(__half);
    else if (out_dtype == 2) TG_SE16(__hip_bfloat16);
    else TG_SE16(float);
#undef TG_SE16
    return check_launch(fn);
  }
  // other sizes and layouts: the step into the next ring slot, then the frames (two launches inside this call)
  if (int rc = tg_step_i8(ring + head_slot * frame_stride_bytes, ring + nxt * frame_stride_bytes, actions, done, overflow, B, S,
                          game_stride_bytes, shift, stream))
    return rc;
  return tg_emit_frames(ring, out, scalars, out_dtype, B, S, T, nxt, t_step, frame_stride_bytes, game_stride_bytes, stream);
}

int tg_gen_from_factors_i8(const int8_t* actions, int8_t* target_out, uint8_t* overflow, int64_t B,
                           int S, int R, int64_t game_stride_bytes, int shift, tg_stream_t stream) {
  if (int rc = validate_common("tg_gen_from_factors_i8", B, S, game_stride_bytes)) return rc;
  if (R < 1 || R > TG_MAX_ACTIONS)
    return fail(TG_ERR_INVALID, "tg_gen_from_factors_i8: R=%d outside [1,%d]", R, TG_MAX_ACTIONS);
  if (B && (!actions || !target_out)) return fail(TG_ERR_INVALID, "tg_gen_from_factors_i8: null pointer");
  tg::ApplyArgs a{nullptr, target_out, actions, nullptr, nullptr, nullptr, overflow, B,
                  game_stride_bytes, game_stride_bytes, S, R, shift};
  return launch_apply<tg::GENF>("tg_gen_from_factors_i8", a, static_cast<hipStream_t>(stream));
}

int tg_copy_i8(const int8_t* state_in, int8_t* state_out, int64_t B, int S, int64_t in_stride_bytes,
               int64_t out_stride_bytes, tg_stream_t stream) {
  if (int rc = validate_common("tg_copy_i8", B, S, in_stride_bytes)) return rc;
  if (int rc = validate_common("tg_copy_i8", B, S, out_stride_bytes)) return rc;
  if (B == 0) return TG_OK;
  if (!state_in || !state_out) return fail(TG_ERR_INVALID, "tg_copy_i8: null pointer");
  if (state_in == state_out) return in_stride_bytes == out_stride_bytes ? TG_OK : fail(TG_ERR_INVALID, "tg_copy_i8: in place with different strides");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int N = S * S * S;
  (void)hipGetLastError();
  if (aligned16(state_in) && aligned16(state_out) && in_stride_bytes % 16 == 0 && out_stride_bytes % 16 == 0) {
    const int nchunk = (N + 15) / 16;
    int sh = -1;
    for (int t = 0; t < 12; ++t)
      if ((1 << t) == nchunk) sh = t;
    const int64_t blocks = (B * nchunk + tg::kBlock - 1) / tg::kBlock;
    if (blocks > 0x7fffffffLL) return fail(TG_ERR_INVALID, "tg_copy_i8: B too large");
    // Out of place, by the bytes both buffers hold together (measured, tg_copy_i8 ping-pong between two buffers, plain /
    // nt loads / nt loads + stores): 256 MiB 33 / 34 / 41 us, 384 MiB 67 / 49 / 61, 512 MiB 88 / 71 / 81, 768 MiB
    // 131 / 127 / 120, 1 GiB 172 / 167 / 160.  This kernel is the bench's copy ceiling: it has to be the best copy.
    const int64_t both = state_in == state_out ? 0 : B * (in_stride_bytes + out_stride_bytes);
    if (TG_SWITCH("TG_COPY_PLAIN"))
      hipLaunchKernelGGL(tg::copy_kernel<0>, dim3((unsigned)blocks), dim3(tg::kBlock), 0, st, state_in, state_out, B, nchunk,
                         sh, N % 16, in_stride_bytes, out_stride_bytes);
    else if (both > (640ll << 20) || TG_SWITCH("TG_COPY_NT2"))
      hipLaunchKernelGGL(tg::copy_kernel<2>, dim3((unsigned)blocks), dim3(tg::kBlock), 0, st, state_in, state_out, B, nchunk,
                         sh, N % 16, in_stride_bytes, out_stride_bytes);
    else if (both > (256ll << 20) || TG_SWITCH("TG_COPY_NT1"))
      hipLaunchKernelGGL(tg::copy_kernel<1>, dim3((unsigned)blocks), dim3(tg::kBlock), 0, st, state_in, state_out, B, nchunk,
                         sh, N % 16, in_stride_bytes, out_stride_bytes);
    else
      hipLaunchKernelGGL(tg::copy_kernel<0>, dim3((unsigned)blocks), dim3(tg::kBlock), 0, st, state_in, state_out, B, nchunk,
                         sh, N % 16, in_stride_bytes, out_stride_bytes);
  } else {
    hipLaunchKernelGGL(tg::copy_bytes_kernel, dim3(capped_grid(B > 65536 ? 65536 : B)), dim3(tg::kBlock), 0, st, state_in,
                       state_out, B, N, in_stride_bytes, out_stride_bytes);
  }
  return check_launch("tg_copy_i8");
}

int tg_done_i8(const int8_t* state, uint8_t* done, int32_t* nnz, int64_t B, int S,
               int64_t game_stride_bytes, tg_stream_t stream) {
  if (int rc = validate_common("tg_done_i8", B, S, game_stride_bytes)) return rc;
  if (B == 0) return TG_OK;
  if (!state || !done) return fail(TG_ERR_INVALID, "tg_done_i8: null pointer");
  const int vec16 = aligned16(state) && game_stride_bytes % 16 == 0;
  const int N = S * S * S;
  int lpg = 1;
  // lanes per game: up to four 16-byte chunks per lane for games of 16 chunks and more (S=9: 16 lanes x 3 chunks, four
  // games per wavefront -- with a wavefront per game 46 lanes did one load each and the launch was latency-bound:
  // 12-16 us for 24 MB), one chunk per lane for the small ones (S=4: 4 lanes)
  while (lpg < 64 && lpg * 16 * (N >= 256 ? 4 : 1) < N) lpg <<= 1;
  const int64_t blocks = (B * lpg + tg::kBlock - 1) / tg::kBlock;
  (void)hipGetLastError(); hipLaunchKernelGGL(tg::done_kernel, dim3(capped_grid(blocks > 8192 ? 8192 : blocks)), dim3(tg::kBlock), 0,
                     static_cast<hipStream_t>(stream), state, done, nnz, B, N, game_stride_bytes, vec16, lpg);
  return check_launch("tg_done_i8");
}

int tg_reset_matmul_i8(int8_t* state_out, int64_t B, int n, int64_t game_stride_bytes,
                       tg_stream_t stream) {
  if (n < 1 || n * n > TG_MAX_S) return fail(TG_ERR_INVALID, "tg_reset_matmul_i8: n=%d, need 1 <= n*n <= %d", n, TG_MAX_S);
  if (int rc = validate_common("tg_reset_matmul_i8", B, n * n, game_stride_bytes)) return rc;
  if (B == 0) return TG_OK;
  if (!state_out) return fail(TG_ERR_INVALID, "tg_reset_matmul_i8: null pointer");
  const int S = n * n, N = S * S * S;
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  hipLaunchKernelGGL(tg::matmul_template_kernel, dim3((N + tg::kBlock - 1) / tg::kBlock), dim3(tg::kBlock), 0, st,
                     state_out, n);
  if (B > 1) {
    const int vec16 = aligned16(state_out) && game_stride_bytes % 16 == 0;
    const int64_t work = (B - 1) * (vec16 ? (N + 15) / 16 : N);
    hipLaunchKernelGGL(tg::broadcast_kernel, dim3(capped_grid((work + tg::kBlock - 1) / tg::kBlock > 8192 ? 8192 : (work + tg::kBlock - 1) / tg::kBlock)),
                       dim3(tg::kBlock), 0, st, state_out, state_out, (int64_t)1, B, N, game_stride_bytes, vec16);
  }
  return check_launch("tg_reset_matmul_i8");
}

int tg_reset_broadcast_i8(const int8_t* start, int8_t* state_out, int64_t B, int S,
                          int64_t game_stride_bytes, tg_stream_t stream) {
  if (int rc = validate_common("tg_reset_broadcast_i8", B, S, game_stride_bytes)) return rc;
  if (B == 0) return TG_OK;
  if (!start || !state_out) return fail(TG_ERR_INVALID, "tg_reset_broadcast_i8: null pointer");
  const int vec16 = aligned16(start) && aligned16(state_out) && game_stride_bytes % 16 == 0;
  const int N = S * S * S;
  const int64_t work = B * (vec16 ? (N + 15) / 16 : N);
  const int64_t blocks = (work + tg::kBlock - 1) / tg::kBlock;
  (void)hipGetLastError(); hipLaunchKernelGGL(tg::broadcast_kernel, dim3(capped_grid(blocks > 8192 ? 8192 : blocks)), dim3(tg::kBlock), 0,
                     static_cast<hipStream_t>(stream), start, state_out, (int64_t)0, B, N, game_stride_bytes, vec16);
  return check_launch("tg_reset_broadcast_i8");
}

}  // extern "C"
