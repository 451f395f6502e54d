// Device-side helpers shared by the tensor-game kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tensor_game.h"

// Measurement switches (environment variables) exist only in the A/B build (mat_mul_amd/build.py, ab=True);
// in the product library TG_SWITCH() is the constant false and no entry point reads the environment.
#ifdef TG_AB_SWITCHES
#include <cstdlib>
#define TG_SWITCH(name) ([]() -> bool { static const bool v = getenv(name) != nullptr; return v; }())
#else
#define TG_SWITCH(name) false
#endif

namespace tg {

constexpr int kBlock = 256;  // 4 wavefronts of 64

// Non-temporal 16-byte store, for pure write streams: the children of tg_expand_i8 at S = 4 and S = 16
// (tools/expand_probe.hip, 2^20 parents x 8 children, 537 MB written: 146 us with plain stores, 97 us with nt stores;
// the product kernels 163 -> 110 us at S = 4, 56.5 -> 50.6 us at S = 16) and the model-input frames (tg_aux.hip).
// Not for S = 25 children (124 -> 134 us: 15 625-byte children end in partial lines) and not for in-place streams (S = 4,
// 4 M games: 78.2 / 78.1 us) -- those gain from non-temporal LOADS instead (s4_kernel, s16_step_kernel, s25_step_kernel).
// ALWAYS select it by a template parameter: behind a run-time flag hipcc merges the two stores into a plain one, and an
// A/B run then measures nothing (that happened to the first round of these experiments).
typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_nt(void* p, const uint4& q) {
  __builtin_nontemporal_store(v4u_t{q.x, q.y, q.z, q.w}, reinterpret_cast<v4u_t*>(p));
}
// S = 4 children of at least this many bytes leave by non-temporal stores (half of the 256 MiB Infinity Cache: below
// it the consumer may still find them there)
constexpr int64_t kStreamOutBytes = 128ll << 20;

// ---- packed int8 <-> int32 ------------------------------------------------------------------

// sign-extended byte t (0..3) of a dword: one v_bfe_i32
__device__ __forceinline__ int sbyte(uint32_t w, int t) {
  return __builtin_amdgcn_sbfe(static_cast<int>(w), 8 * t, 8);
}

// low bytes of four ints -> one dword (3 x v_perm_b32)
__device__ __forceinline__ uint32_t pack4(int n0, int n1, int n2, int n3) {
  uint32_t lo = __builtin_amdgcn_perm(static_cast<uint32_t>(n1), static_cast<uint32_t>(n0), 0x0c0c0400u);
  uint32_t hi = __builtin_amdgcn_perm(static_cast<uint32_t>(n3), static_cast<uint32_t>(n2), 0x0c0c0400u);
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// a*b for |a|,|b| < 2^23 as ONE full-rate v_mul_i32_i24.  hipcc (ROCm 7.2) lowers __mul24 of
// values whose range it cannot see to the quarter-rate v_mul_lo_u32; the asm pins the opcode.
__device__ __forceinline__ int mul24_pinned(int a, int b) {
  int r;
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// a*b + c, same operand range, as ONE v_mad_i32_i24 (low 32 bits of the exact result)
__device__ __forceinline__ int mad24_pinned(int a, int b, int c) {
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// the same with a wave-uniform first factor kept in an SGPR (one scalar operand per VOP3 is allowed)
__device__ __forceinline__ int mad24_sgpr(int a_uniform, int b, int c) {
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "s"(a_uniform), "v"(b), "v"(c));
  return r;
}

// number of non-zero bytes of a dword, accumulated: TWO instructions.  v_msad_u8 sums |a_i - b_i| over the bytes with
// b_i != 0, and x ^ 0x01010101 differs from x by exactly one in every byte (round 3; the mask-and-popcount form took four)
__device__ __forceinline__ int count_nonzero_bytes(uint32_t x, int acc = 0) {
  return static_cast<int>(__builtin_amdgcn_msad_u8(x ^ 0x01010101u, x, static_cast<uint32_t>(acc)));
}
__device__ __forceinline__ int count_nonzero_bytes(const uint4& q) {
  return count_nonzero_bytes(q.x, count_nonzero_bytes(q.y, count_nonzero_bytes(q.z, count_nonzero_bytes(q.w))));
}

// ---- the 64-bit state key (tg_hash_u64, include/tensor_game.h) -------------------------------
// With the S^3 bytes zero-padded to 8-byte little-endian words w_k:
//   H = fmix64( (sum_k fmix64(w_k + (k+1) * 0x9E3779B97F4A7C15)) ^ (S^3 * 0xC2B2AE3D27D4EB4F) ),
// fmix64 = the MurmurHash3 finaliser.  The sum is order independent, so any mapping of chunks to lanes is valid.
__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xFF51AFD7ED558CCDull;
  k ^= k >> 33;
  k *= 0xC4CEB9FE1A85EC53ull;
  k ^= k >> 33;
  return k;
}
// the contribution of the 16-byte chunk c (words 2c and 2c+1) of a game
__device__ __forceinline__ uint64_t hash_chunk(const uint4& q, int c) {
  const uint64_t w0 = static_cast<uint64_t>(q.x) | (static_cast<uint64_t>(q.y) << 32);
  const uint64_t w1 = static_cast<uint64_t>(q.z) | (static_cast<uint64_t>(q.w) << 32);
  return fmix64(w0 + static_cast<uint64_t>(2 * c + 1) * 0x9E3779B97F4A7C15ull) +
         fmix64(w1 + static_cast<uint64_t>(2 * c + 2) * 0x9E3779B97F4A7C15ull);
}
__device__ __forceinline__ uint64_t hash_finish(uint64_t sum, int N) {
  return fmix64(sum ^ (static_cast<uint64_t>(N) * 0xC2B2AE3D27D4EB4Full));
}

// ---- team (sub-wave) reductions --------------------------------------------------------------

// OR-reduce a predicate over the TS consecutive lanes (TS a power of two <= 64) this lane belongs to.
template <int TS>
__device__ __forceinline__ bool team_any(bool p) {
  const uint64_t m = __ballot(p);
  if constexpr (TS >= 64) {
    return m != 0;
  } else {
    const int lane = threadIdx.x & 63;
    const int base = lane & ~(TS - 1);
    return ((m >> base) & ((1ull << TS) - 1ull)) != 0;
  }
}

// ---- packed int16 helpers used by several kernel families ------------------------------------
__device__ __forceinline__ uint32_t pk_sub_u16_sat(uint32_t a, uint32_t b) {  // max(a - b, 0) per half
  uint32_t d;
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) {  // low 16 bits of a*b + c per half
  uint32_t d;
  asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// ---- wavefront scans / sums, LDS-only barrier ---------------------------------------------------
// inclusive prefix sum over the 64 lanes of a wavefront, all in the VALU (DPP row shifts and row broadcasts)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
  x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0u, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0u, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return x;
}
// the wavefront's total of a per-lane integer (uniform)
__device__ __forceinline__ int wave_sum(int x) {
  return __builtin_amdgcn_readlane(static_cast<int>(wave_inclusive_scan(static_cast<uint32_t>(x))), 63);
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is a workgroup-scope fence + s_barrier: hipcc puts
// `s_waitcnt vmcnt(0)` in front of it, i.e. every barrier also waits until all global STORES this wavefront has issued
// are acknowledged -- for a kernel that streams results out between barriers and never reads them back that drains
// the write queue (microseconds under load) once per phase.  Use this one where the data exchanged across the barrier
// lives in LDS and nothing written to global memory is read again by the workgroup.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- the factor distribution of the generator (host-filled, passed by value) ----------------
// The basis sampler draws 32-bit uniforms against `thr`.  The factor generator draws SIXTEEN-bit uniforms,
// eight per Philox block: a draw d16 selects val[#{t : d16 * 2^16 >= thr[t]}], i.e. d16 is compared with
// thr16[t] = ceil(thr[t] / 2^16) in [0, 65536].  For the packed evaluation (two draws per dword) the
// thresholds that are always true (thr16 == 0) are folded into `base16` and those never true (65536) are
// dropped: value = base + sum_{t < nthr} [d16 >= c16[t] + 1] * delta16[t], all int16 replicated in both halves.
struct Dist {
  uint32_t thr[TG_MAX_VALUES - 1];
  int8_t val[TG_MAX_VALUES];
  int nv;
  uint32_t thr16[TG_MAX_VALUES - 1];
  int nthr;
  uint32_t c16[TG_MAX_VALUES - 1];
  uint32_t delta16[TG_MAX_VALUES - 1];
  uint32_t base16;
};

// one 16-bit draw -> value (scalar form; gen_tokens_kernel)
__device__ __forceinline__ int draw_value16(uint32_t d16, const Dist& D) {
  if (D.nv == 3) {  // the reference's vocabulary (-1,0,1): two compares, two selects (wave-uniform branch)
    const int v = d16 >= D.thr16[0] ? D.val[1] : D.val[0];
    return d16 >= D.thr16[1] ? D.val[2] : v;
  }
  int idx = 0;
#pragma unroll
  for (int t = 0; t < TG_MAX_VALUES - 1; ++t) idx += (t < D.nv - 1) && (d16 >= D.thr16[t]);
  int v = D.val[0];
#pragma unroll
  for (int t = 1; t < TG_MAX_VALUES; ++t) v = (idx == t) ? D.val[t] : v;
  return v;
}

// Packed evaluation of the draws of one Philox block: each output word holds two 16-bit draws; per threshold
// m = min(max(d - (thr16 - 1), 0), 1) = [d >= thr16] and value += m * delta: three packed ops for two draws.  The
// distribution's constants are wave-uniform and enter as SGPR operands (one per VOP3P instruction); `one` = 0x00010001
// and `base` = D.base16 live in VGPRs of the caller.
__device__ __forceinline__ uint32_t draw_step16(uint32_t w, uint32_t c16, uint32_t delta16, uint32_t one, uint32_t v) {
  uint32_t m;
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(m) : "v"(w), "s"(c16));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(m), "v"(one));
  asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(v) : "v"(m), "s"(delta16), "v"(v));
  return v;
}

// eight draws (one Philox block) -> four dwords of two int16 values each: P[m] = values of elements 2m, 2m+1
// TERNARY = true: the caller knows D.nthr == 2 (host-checked): the general form is not even compiled in, which keeps its
// fourteen threshold / delta scalars out of the SGPR budget of the fused generator.
template <bool TERNARY = false>
__device__ __forceinline__ void draw_block16(const uint32_t (&w)[4], const Dist& D, uint32_t one, uint32_t base,
                                             uint32_t (&P)[4]) {
  if (TERNARY || D.nthr == 2) {  // wave-uniform: the ternary vocabulary of the reference.  ONE asm statement: the four words are
    // interleaved (independent chains) and no compiler padding separates the dependent packed ops
    uint32_t t0, t1, t2, t3;
    asm("v_pk_sub_u16 %4, %8, %12 clamp\n\tv_pk_sub_u16 %5, %9, %12 clamp\n\t"
        "v_pk_sub_u16 %6, %10, %12 clamp\n\tv_pk_sub_u16 %7, %11, %12 clamp\n\t"
        "v_pk_min_u16 %4, %4, %16\n\tv_pk_min_u16 %5, %5, %16\n\tv_pk_min_u16 %6, %6, %16\n\tv_pk_min_u16 %7, %7, %16\n\t"
        "v_pk_mad_u16 %0, %4, %14, %17\n\tv_pk_mad_u16 %1, %5, %14, %17\n\t"
        "v_pk_mad_u16 %2, %6, %14, %17\n\tv_pk_mad_u16 %3, %7, %14, %17\n\t"
        "v_pk_sub_u16 %4, %8, %13 clamp\n\tv_pk_sub_u16 %5, %9, %13 clamp\n\t"
        "v_pk_sub_u16 %6, %10, %13 clamp\n\tv_pk_sub_u16 %7, %11, %13 clamp\n\t"
        "v_pk_min_u16 %4, %4, %16\n\tv_pk_min_u16 %5, %5, %16\n\tv_pk_min_u16 %6, %6, %16\n\tv_pk_min_u16 %7, %7, %16\n\t"
        "v_pk_mad_u16 %0, %4, %15, %0\n\tv_pk_mad_u16 %1, %5, %15, %1\n\t"
        "v_pk_mad_u16 %2, %6, %15, %2\n\tv_pk_mad_u16 %3, %7, %15, %3"
        : "=&v"(P[0]), "=&v"(P[1]), "=&v"(P[2]), "=&v"(P[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "s"(D.c16[0]), "s"(D.c16[1]), "s"(D.delta16[0]), "s"(D.delta16[1]),
          "v"(one), "v"(base));
    return;
  }
  if constexpr (TERNARY) return;
#pragma unroll
  for (int m = 0; m < 4; ++m) P[m] = base;
#pragma unroll
  for (int t = 0; t < TG_MAX_VALUES - 1; ++t) {
    if (t < D.nthr) {  // wave-uniform
#pragma unroll
      for (int m = 0; m < 4; ++m) P[m] = draw_step16(w[m], D.c16[t], D.delta16[t], one, P[m]);
    }
  }
}

// ---- Philox-4x32-10 (Salmon et al. SC'11); bit-identical to oracle/tensor_game.py -----------

struct U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one v_mad_u64_u32 per product (hi and lo together) instead of v_mul_hi_u32 + v_mul_lo_u32
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c.x;
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c.z;
    c = U4{static_cast<uint32_t>(p1 >> 32) ^ c.y ^ k0, static_cast<uint32_t>(p1),
           static_cast<uint32_t>(p0 >> 32) ^ c.w ^ k1, static_cast<uint32_t>(p0)};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

}  // namespace tg
