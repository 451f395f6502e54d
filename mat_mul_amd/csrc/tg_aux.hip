// "Next rows" of the hot path: model-input assembly (N1), state hashing
// (N2) and the exact slice-rank reward (N3).  gfx950 only; part of libtensorgame.so.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>

#include "../../include/tensor_game.h"
#include "tg_device.h"

int tg_internal_fail(int code, const char* fmt, ...);  // tg_kernels.hip

namespace tg {

// 16 output bytes from PER = 16 / sizeof(OutT) int8 values (small integers: exact in float32, float16 and bfloat16
// alike).  Built in registers, word by word: an `OutT v[PER]` array + memcpy made hipcc stage the values through LDS,
// and the f16 / bf16 kernels took 33 us where the f32 kernel took 13.5 (S=4, B=65 536, T=4).
template <typename OutT>
__device__ __forceinline__ uint4 emit_pack(const int (&x)[16 / sizeof(OutT)]) {
  if constexpr (sizeof(OutT) == 4) {
    return uint4{__float_as_uint(static_cast<float>(x[0])), __float_as_uint(static_cast<float>(x[1])),
                 __float_as_uint(static_cast<float>(x[2])), __float_as_uint(static_cast<float>(x[3]))};
  } else {
    uint32_t w[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float lo = static_cast<float>(x[2 * d]), hi = static_cast<float>(x[2 * d + 1]);
      if constexpr (std::is_same<OutT, __half>::value) {
        typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
        const h2_t h = __builtin_amdgcn_cvt_pkrtz(lo, hi);  // |x| <= 128: exact whatever the rounding
        __builtin_memcpy(&w[d], &h, 4);
      } else {  // bfloat16 = the upper half of the float32 (|x| <= 128 has at most 8 significant bits: exact)
        w[d] = __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
      }
    }
    return uint4{w[0], w[1], w[2], w[3]};
  }
}
// element t of a packed group (the last, partial group of the output)
template <typename OutT>
__device__ __forceinline__ void emit_store_one(OutT* out, const uint4& o, int t) {
  const uint32_t w[4] = {o.x, o.y, o.z, o.w};
  if constexpr (sizeof(OutT) == 4) *reinterpret_cast<uint32_t*>(out) = w[t];
  else *reinterpret_cast<uint16_t*>(out) = static_cast<uint16_t>(w[t >> 1] >> (16 * (t & 1)));
}

// ---------------------------------------------------------------------------------------------
// N1: int8 history ring -> float model input.  One thread per 16 input bytes (64 or 32 output
// bytes); HBM-bound on the float write: S^3*T*(1 + 4) bytes per game for float32.
// ---------------------------------------------------------------------------------------------
template <typename OutT, bool NT>
__global__ __launch_bounds__(kBlock) void emit_frames_kernel(const int8_t* ring, OutT* out, float* scalars,
                                                             int64_t B, int N, int T, int head_slot, float t_step,
                                                             int64_t frame_stride, int64_t game_stride, int vec) {
  // The output (B,T,N) is one flat, 16-byte-aligned array: one thread per 16 OUTPUT bytes (PER = 4
  // floats or 8 halves) of that flat array, so a wavefront always stores one contiguous KiB whatever
  // N is.  A group whose PER source bytes lie in one frame and are PER-aligned is one dword/dwordx2
  // load (vec); otherwise (odd N: groups straddle frames) the source bytes are gathered one by one.
  constexpr int PER = 16 / sizeof(OutT);
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t nthr = static_cast<int64_t>(gridDim.x) * kBlock;
  if (scalars)
    for (int64_t b = tid; b < B; b += nthr) scalars[b] = t_step;
  const int64_t total = B * T * N;  // output elements
  for (int64_t q = tid * PER; q < total; q += nthr * PER) {
    const int64_t bf = q / N;  // b*T + f of the first element
    const int e0 = static_cast<int>(q - bf * N);
    auto frame_ptr = [&](int64_t bfi) {
      const int64_t b = bfi / T;
      const int f = static_cast<int>(bfi - b * T);
      int slot = head_slot - f;
      if (slot < 0) slot += T;
      return ring + b * game_stride + slot * frame_stride;
    };
    const int8_t* src = frame_ptr(bf);
    int v[PER];
    if (vec && e0 + PER <= N) {
      // the group lies inside one frame; frames are 4-byte aligned (vec), the group need not be:
      // aligned dwords + v_alignbyte.  The extra dword is only touched when the group is misaligned,
      // and then it still starts below 4*ceil(N/4) <= frame stride.
      const uint32_t* a = reinterpret_cast<const uint32_t*>(src + (e0 & ~3));
      const uint32_t sh = static_cast<uint32_t>(e0 & 3);
      uint32_t w[PER / 4];
      uint32_t lo = a[0];
#pragma unroll
      for (int d = 0; d < PER / 4; ++d) {
        const bool need_hi = sh != 0 || d + 1 < PER / 4;
        const uint32_t hi = need_hi ? a[d + 1] : 0u;
        w[d] = __builtin_amdgcn_alignbyte(hi, lo, sh);
        lo = hi;
      }
#pragma unroll
      for (int t = 0; t < PER; ++t) v[t] = sbyte(w[t >> 2], t & 3);
    } else {
      const int8_t* nxt = (e0 + PER > N && bf + 1 < B * T) ? frame_ptr(bf + 1) : src;
#pragma unroll
      for (int t = 0; t < PER; ++t) {
        const int e = e0 + t;
        v[t] = (q + t < total) ? (e < N ? src[e] : nxt[e - N]) : 0;
      }
    }
    const uint4 o = emit_pack<OutT>(v);
    if (q + PER <= total) {
      // NT: the model input is a pure write stream -- non-temporal stores (a template parameter: behind a run-time flag
      // hipcc merges the two stores into a plain one).  f32, T=4: 14.1 -> 11.4 us at S=4 B=65 536, 262 -> 227 us at 2^20
      // games, 119 -> 96 us at S=16 B=8 192, 246 -> 185 us at S=25 B=4 096; f16: 139 -> 131, 70 -> 58, 137 -> 112.
      if constexpr (NT) store16_nt(out + q, o);
      else *reinterpret_cast<uint4*>(out + q) = o;
    } else {
#pragma unroll
      for (int t = 0; t < PER; ++t)
        if (q + t < total) emit_store_one(out + q + t, o, t);
    }
  }
}

// x / d for x < 2^31 by multiplication: m = ceil(2^32 / d) (d >= 2) gives floor(x / d) or one more
__device__ __forceinline__ uint32_t udiv_magic(uint32_t x, uint32_t d, uint32_t m) {
  uint32_t q = __umulhi(x, m);
  if (q * d > x) --q;
  return q;
}

// The same kernel for outputs below 2^31 elements and 4-byte aligned frames (every case the env produces):
// 32-bit indices and two divisions by multiplication per group instead of two 64-bit divisions -- the index
// arithmetic was 380 vector instructions (48 of them quarter-rate multiplies) for 16 output bytes.
template <typename OutT, bool NT>
__global__ __launch_bounds__(kBlock) void emit_frames_fast_kernel(const int8_t* ring, OutT* out, float* scalars, int B, int N,
                                                                  int T, int head_slot, float t_step, int64_t frame_stride,
                                                                  int64_t game_stride, uint32_t mN, uint32_t mT) {
  constexpr uint32_t PER = 16 / sizeof(OutT);
  const uint32_t tid = blockIdx.x * kBlock + threadIdx.x, nthr = gridDim.x * kBlock;
  if (scalars)
    for (uint32_t b = tid; b < static_cast<uint32_t>(B); b += nthr) scalars[b] = t_step;
  const uint32_t total = static_cast<uint32_t>(B) * T * N;
  auto frame_ptr = [&](uint32_t bf) {
    const uint32_t b = T > 1 ? udiv_magic(bf, T, mT) : bf;
    const int f = static_cast<int>(bf - b * T);
    int slot = head_slot - f;
    if (slot < 0) slot += T;
    return ring + b * game_stride + slot * frame_stride;
  };
  for (uint32_t q = tid * PER; q < total; q += nthr * PER) {
    const uint32_t bf = udiv_magic(q, N, mN);
    const int e0 = static_cast<int>(q - bf * N);
    const int8_t* src = frame_ptr(bf);
    int v[PER];
    if (e0 + static_cast<int>(PER) <= N) {
      const uint32_t* a = reinterpret_cast<const uint32_t*>(src + (e0 & ~3));
      const uint32_t sh = static_cast<uint32_t>(e0 & 3);
      uint32_t w[PER / 4];
      uint32_t lo = a[0];
#pragma unroll
      for (uint32_t d = 0; d < PER / 4; ++d) {
        const bool need_hi = sh != 0 || d + 1 < PER / 4;
        const uint32_t hi = need_hi ? a[d + 1] : 0u;
        w[d] = __builtin_amdgcn_alignbyte(hi, lo, sh);
        lo = hi;
      }
#pragma unroll
      for (uint32_t t = 0; t < PER; ++t) v[t] = sbyte(w[t >> 2], t & 3);
    } else {  // the group straddles two frames (once per frame when N is not a multiple of PER)
      const int8_t* nxt = (bf + 1 < static_cast<uint32_t>(B) * T) ? frame_ptr(bf + 1) : src;
#pragma unroll
      for (uint32_t t = 0; t < PER; ++t) {
        const int e = e0 + static_cast<int>(t);
        v[t] = (q + t < total) ? (e < N ? src[e] : nxt[e - N]) : 0;
      }
    }
    const uint4 o = emit_pack<OutT>(v);
    if (q + PER <= total) {
      // NT: the model input is a pure write stream -- non-temporal stores (a template parameter: behind a run-time flag
      // hipcc merges the two stores into a plain one).  f32, T=4: 14.1 -> 11.4 us at S=4 B=65 536, 262 -> 227 us at 2^20
      // games, 119 -> 96 us at S=16 B=8 192, 246 -> 185 us at S=25 B=4 096; f16: 139 -> 131, 70 -> 58, 137 -> 112.
      if constexpr (NT) store16_nt(out + q, o);
      else *reinterpret_cast<uint4*>(out + q) = o;
    } else {
#pragma unroll
      for (uint32_t t = 0; t < PER; ++t)
        if (q + t < total) emit_store_one(out + q + t, o, static_cast<int>(t));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// N2: 64-bit state hash.  Team of lpg lanes per game (as done_kernel), 16 bytes = two words per
// lane-iteration, wrapping sum across the team (order independent => any lane mapping is valid).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void hash_kernel(const int8_t* state, uint64_t* out, int64_t B, int N,
                                                      int64_t stride, int vec16, int lpg) {
  const int lt = threadIdx.x & (lpg - 1);
  const int64_t team = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / lpg;
  const int64_t nteam = (static_cast<int64_t>(gridDim.x) * kBlock) / lpg;
  const int64_t rounds = (B + nteam - 1) / nteam;
  const int nword = (N + 7) >> 3;
  for (int64_t it = 0; it < rounds; ++it) {
    const int64_t g = team + it * nteam;
    const bool live = g < B;
    const int8_t* p = state + (live ? g : B - 1) * stride;
    uint64_t h = 0;
    int body_words = 0;
    if (vec16) {
      body_words = (N >> 4) << 1;  // whole 16-byte chunks
      auto mix = [&](const uint4& q, int c) { h += hash_chunk(q, c); };
      const int nc = N >> 4;  // whole chunks
      int c = lt;
      for (; c + 3 * lpg < nc; c += 4 * lpg) {  // four chunks in flight per lane
        const uint4 q0 = *reinterpret_cast<const uint4*>(p + 16 * c), q1 = *reinterpret_cast<const uint4*>(p + 16 * (c + lpg)),
                    q2 = *reinterpret_cast<const uint4*>(p + 16 * (c + 2 * lpg)), q3 = *reinterpret_cast<const uint4*>(p + 16 * (c + 3 * lpg));
        mix(q0, c);
        mix(q1, c + lpg);
        mix(q2, c + 2 * lpg);
        mix(q3, c + 3 * lpg);
      }
      for (; c < nc; c += lpg) mix(*reinterpret_cast<const uint4*>(p + 16 * c), c);
    }
    for (int k = body_words + lt; k < nword; k += lpg) {  // remaining words, byte by byte, zero padded
      uint64_t w = 0;
      for (int t = 0; t < 8; ++t) {
        const int e = 8 * k + t;
        if (e < N) w |= static_cast<uint64_t>(static_cast<uint8_t>(p[e])) << (8 * t);
      }
      h += fmix64(w + static_cast<uint64_t>(k + 1) * 0x9E3779B97F4A7C15ull);
    }
    for (int off = lpg >> 1; off > 0; off >>= 1) h += __shfl_xor(h, off);
    if (lt == 0 && live) out[g] = hash_finish(h, N);
  }
}

// ---------------------------------------------------------------------------------------------
// N2, second half: membership of 64-bit keys in the transposition table (act.py:188-195: `c not in new_mc_tree`;
// act.py:209-211: the expanded state's key enters the tree).  The table is a caller-owned open-addressing array of
// uint64 (capacity a power of two, 0 = empty slot; a key that IS 0 is stored as kSeenZeroKey), linear probing from
// key & (capacity - 1) -- the keys are fmix64 outputs, already uniform.  Lookup and insertion are TWO kernels of one
// call, so `fresh` is decided against the table as it was before the call (the reference filters all candidates of
// an expansion first and records keys afterwards), independent of how lanes interleave.
// ---------------------------------------------------------------------------------------------
constexpr uint64_t kSeenZeroKey = 0x9E3779B97F4A7C15ull;

__global__ __launch_bounds__(kBlock) void seen_lookup_kernel(const uint64_t* keys, const uint64_t* table, uint64_t capmask,
                                                             uint8_t* fresh, const uint8_t* mask, int64_t n) {
  const int64_t nthr = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += nthr) {
    if (mask && !mask[i]) {
      fresh[i] = 0;
      continue;
    }
    uint64_t k = keys[i];
    if (k == 0) k = kSeenZeroKey;
    uint64_t slot = k & capmask;
    bool found = false;
    for (uint64_t probes = 0; probes <= capmask; ++probes) {  // a full table ends the walk after capacity probes
      const uint64_t t = table[slot];
      if (t == k) {
        found = true;
        break;
      }
      if (t == 0) break;
      slot = (slot + 1) & capmask;
    }
    fresh[i] = found ? 0 : 1;
  }
}

__global__ __launch_bounds__(kBlock) void seen_insert_kernel(const uint64_t* keys, uint64_t* table, uint64_t capmask,
                                                             const uint8_t* mask, int64_t n, uint32_t* status) {
  const int64_t nthr = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += nthr) {
    if (mask && !mask[i]) continue;
    uint64_t k = keys[i];
    if (k == 0) k = kSeenZeroKey;
    uint64_t slot = k & capmask;
    bool placed = false;
    for (uint64_t probes = 0; probes <= capmask; ++probes) {
      const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(table + slot), 0ull,
                                               static_cast<unsigned long long>(k));
      if (old == 0ull || old == k) {  // claimed the empty slot, or the key is there already (possibly a sibling's)
        placed = true;
        break;
      }
      slot = (slot + 1) & capmask;
    }
    if (!placed && status) atomicOr(status, 1u);  // table full: the key was NOT recorded
  }
}

// ---------------------------------------------------------------------------------------------
// N3: sum of slice ranks.  One wavefront per (game, slice i): lane r holds row r of the S x S
// matrix state[b][i] reduced mod p; S elimination steps, each a ballot (pivot search), a broadcast
// of the pivot row (shuffle) and a cross-multiplied update (no modular inverse):
// row_r <- row_r * piv_c - row_p * row_r[c]  (mod p).  Two primes, one per half-wave, ranks maxed.
//
// The arithmetic is done in DOUBLE PRECISION on balanced residues |x| <= p/2 with p < 2^26 (round 2; round 1 used
// 31-bit primes and 64-bit integer products, ~26 integer instructions per element): t = a*pc - pk*mine is exact
// (|t| < 2^51), q = rint(t / p) is off by less than 2^-24 before rounding, r = fma(-q, p, t) is exact and lands in
// [-p/2, p/2] again -- five fp64 instructions per element, and CDNA4 issues v_fma_f64 at the rate of any other VOP3
// (tools/issue_rate_probe.hip).  "Is this residue zero" is an exact comparison with 0.0.  A rank mod p can only be
// too small, and is so with probability ~S/p per slice: both primes wrong on the same slice ~1e-13.
// ---------------------------------------------------------------------------------------------
// Rank of one S x S slice modulo BOTH primes at once: lanes 0..31 eliminate mod 2^26 - 5, lanes 32..63 the
// same matrix mod 2^26 - 27 (S <= 32 rows per half-wave); the halves choose their own pivots.  Returns the
// larger of the two ranks.
// W = rows per group (a power of two >= S, <= 32): a half-wave holds 32 / W slices side by side (S = 16: two,
// S = 9: two, S = 4: eight), each group with its own pivots -- the lanes of a wavefront were 50 % idle at S = 16 and
// 87 % at S = 4 with one slice per half.  Returns the sum of the ranks of the slices i0 .. i0 + 32 / W - 1 (those < S).
template <int ST, int W>
__device__ __forceinline__ int slice_rank2(const int8_t* game, int S, int i0, int lane) {
  constexpr int SMAX = ST ? ST : TG_MAX_S;
  constexpr int GROUPS = 32 / W;
  const bool upper = lane >= 32;
  const int r0 = lane & (W - 1), grp = (lane & 31) / W;  // row within the slice, slice within the pass
  const int slice = i0 + grp;
  const bool mine_valid = r0 < S && slice < S;
  const int8_t* m = game + (slice < S ? slice : 0) * S * S;
  const double P = upper ? 67108837.0 : 67108859.0;
  const double invP = upper ? (1.0 / 67108837.0) : (1.0 / 67108859.0);
  double row[SMAX];
#pragma unroll
  for (int c = 0; c < SMAX; ++c) row[c] = (mine_valid && c < S) ? static_cast<double>(m[r0 * S + c]) : 0.0;
  bool used = !mine_valid;  // rows already chosen as pivots (and the idle lanes)
  int rank = 0;
  const int gbase = lane & ~(W - 1);  // first lane of my group
#pragma unroll
  for (int c = 0; c < SMAX; ++c) {
    if (c < S) {
      const uint64_t cand = __ballot(!used && row[c] != 0.0);
      if (cand) {  // wave-uniform; a group without a candidate just idles through the step
        const uint32_t gmask = static_cast<uint32_t>(cand >> gbase) & (W == 32 ? 0xFFFFFFFFu : ((1u << W) - 1u));
        const bool has = gmask != 0;
        const int pr = gbase + (has ? __builtin_ctz(gmask) : 0);  // this group's pivot row
        const double pc = __shfl(row[c], pr);
        const double mine = row[c];
        const bool upd = has && !used && lane != pr && mine != 0.0;
        // columns < c of every unused row were zeroed by earlier pivots, and column c becomes zero
#pragma unroll
        for (int k = c + 1; k < SMAX; ++k) {
          if (k < S) {
            const double pk = __shfl(row[k], pr);
            if (upd) {
              const double t = __builtin_fma(row[k], pc, -(pk * mine));
              const double q = __builtin_rint(t * invP);
              row[k] = __builtin_fma(-q, P, t);
            }
          }
        }
        if (upd) row[c] = 0.0;
        if (has && lane == pr) used = true;
        rank += has;
      }
    }
  }
  int total = 0;
#pragma unroll
  for (int q = 0; q < GROUPS; ++q) {  // per slice: the larger of the two primes' ranks
    const int r1 = __builtin_amdgcn_readlane(rank, q * W), r2 = __builtin_amdgcn_readlane(rank, 32 + q * W);
    total += r1 > r2 ? r1 : r2;
  }
  return total;
}

template <int ST>
__global__ __launch_bounds__(kBlock) void rank_kernel(const int8_t* state, int32_t* out, int64_t B, int Srt,
                                                      int64_t stride) {
  __shared__ int partial[kBlock / 64];
  constexpr int W = ST == 0 ? 32 : (ST <= 4 ? 4 : (ST <= 8 ? 8 : (ST <= 16 ? 16 : 32)));
  constexpr int GROUPS = 32 / W;
  const int S = ST ? ST : Srt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (ST != 0 && ST <= GROUPS) {  // S = 4: one pass of one wavefront is a whole game -- a wavefront per game
    for (int64_t g = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave; g < B; g += static_cast<int64_t>(gridDim.x) * (kBlock / 64)) {
      const int t = slice_rank2<ST, W>(state + g * stride, S, 0, lane);
      if (lane == 0) out[g] = t;
    }
    return;
  }
  // one workgroup per game: its 4 wavefronts walk the S slices, 32 / W at a time
  for (int64_t g = blockIdx.x; g < B; g += gridDim.x) {
    int acc = 0;
    for (int i = wave * GROUPS; i < S; i += (kBlock / 64) * GROUPS) acc += slice_rank2<ST, W>(state + g * stride, S, i, lane);
    if (lane == 0) partial[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < kBlock / 64; ++w) t += partial[w];
      out[g] = t;
    }
    __syncthreads();
  }
}

}  // namespace tg

namespace {
int launched(const char* fn) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tg_internal_fail(TG_ERR_HIP, "%s: %s", fn, hipGetErrorString(e));
  return TG_OK;
}
unsigned grid_for(int64_t blocks, int64_t cap) {
  return static_cast<unsigned>(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}
int check_state(const char* fn, int64_t B, int S, int64_t stride) {
  if (B < 0 || S < 1 || S > TG_MAX_S || stride < (int64_t)S * S * S)
    return tg_internal_fail(TG_ERR_INVALID, "%s: bad B/S/stride", fn);
  return TG_OK;
}
}  // namespace

extern "C" {

int tg_emit_frames(const int8_t* ring, void* out, float* scalars, int out_dtype, int64_t B, int S,
                   int T, int head_slot, float t_step, int64_t frame_stride_bytes,
                   int64_t game_stride_bytes, tg_stream_t stream) {
  const char* fn = "tg_emit_frames";
  if (int rc = check_state(fn, B, S, frame_stride_bytes)) return rc;
  if (T < 1 || T > 64 || head_slot < 0 || head_slot >= T)
    return tg_internal_fail(TG_ERR_INVALID, "%s: need 1 <= T <= 64 and 0 <= head_slot < T", fn);
  if (game_stride_bytes < (int64_t)(T - 1) * frame_stride_bytes + (int64_t)S * S * S)
    return tg_internal_fail(TG_ERR_INVALID, "%s: game_stride_bytes too small for T frames", fn);
  if (B == 0) return TG_OK;
  if (!ring || !out) return tg_internal_fail(TG_ERR_INVALID, "%s: null pointer", fn);
  const int N = S * S * S;
  if (reinterpret_cast<uintptr_t>(out) & 15) return tg_internal_fail(TG_ERR_INVALID, "%s: out must be 16-byte aligned", fn);
  // dword loads need 4-byte aligned frames
  if (out_dtype < 0 || out_dtype > 2) return tg_internal_fail(TG_ERR_INVALID, "%s: out_dtype must be 0 (f32), 1 (f16) or 2 (bf16)", fn);
  const int per = out_dtype ? 8 : 4;
  const int vec16 = (reinterpret_cast<uintptr_t>(ring) % 4) == 0 && frame_stride_bytes % 4 == 0 &&
                    game_stride_bytes % 4 == 0;
  const int64_t work = (B * T * N + per - 1) / per;
  const dim3 grid(grid_for((work + tg::kBlock - 1) / tg::kBlock, 32768)), block(tg::kBlock);
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  const int64_t total = B * T * N;
  // outputs the Infinity Cache can hold stay there for their reader (tools measurement on expand's children: nt stores
  // of a 34 MB output speed the producer up and slow the consumer down by the same amount)
  const bool nt = total * (out_dtype ? 2 : 4) >= tg::kStreamOutBytes || TG_SWITCH("TG_EMIT_NT");  // (A/B switch: tests)
  if (vec16 && N >= 2 && total < (1ll << 31)) {
    // (the 32-bit loop variable passes `total` by at most one grid stride, 2^26 elements: no wrap)
    const uint32_t mN = static_cast<uint32_t>(((1ull << 32) + N - 1) / N);
    const uint32_t mT = T > 1 ? static_cast<uint32_t>(((1ull << 32) + T - 1) / T) : 0u;
#define TG_EMIT_FAST(OutT_, NT_)                                                                                        \
  hipLaunchKernelGGL((tg::emit_frames_fast_kernel<OutT_, NT_>), grid, block, 0, st, ring, static_cast<OutT_*>(out), scalars, \
                     static_cast<int>(B), N, T, head_slot, t_step, frame_stride_bytes, game_stride_bytes, mN, mT)
    if (out_dtype == 1) { if (nt) TG_EMIT_FAST(__half, true); else TG_EMIT_FAST(__half, false); }
    else if (out_dtype == 2) { if (nt) TG_EMIT_FAST(__hip_bfloat16, true); else TG_EMIT_FAST(__hip_bfloat16, false); }
    else { if (nt) TG_EMIT_FAST(float, true); else TG_EMIT_FAST(float, false); }
#undef TG_EMIT_FAST
    return launched(fn);
  }
#define TG_EMIT(OutT_, NT_)                                                                                              \
  hipLaunchKernelGGL((tg::emit_frames_kernel<OutT_, NT_>), grid, block, 0, st, ring, static_cast<OutT_*>(out), scalars, B, N, \
                     T, head_slot, t_step, frame_stride_bytes, game_stride_bytes, vec16)
  if (out_dtype == 1) { if (nt) TG_EMIT(__half, true); else TG_EMIT(__half, false); }
  else if (out_dtype == 2) { if (nt) TG_EMIT(__hip_bfloat16, true); else TG_EMIT(__hip_bfloat16, false); }
  else { if (nt) TG_EMIT(float, true); else TG_EMIT(float, false); }
#undef TG_EMIT
  return launched(fn);
}

// the key pass of tg_expand_keyed_i8 for the kernel families that do not produce the keys themselves (tg_kernels.hip)
int tg_internal_hash(const int8_t* state, uint64_t* hash_out, int64_t B, int S, int64_t stride, hipStream_t st) {
  return tg_hash_u64(state, hash_out, B, S, stride, st);
}

int tg_hash_u64(const int8_t* state, uint64_t* hash_out, int64_t B, int S, int64_t game_stride_bytes,
                tg_stream_t stream) {
  const char* fn = "tg_hash_u64";
  if (int rc = check_state(fn, B, S, game_stride_bytes)) return rc;
  if (B == 0) return TG_OK;
  if (!state || !hash_out) return tg_internal_fail(TG_ERR_INVALID, "%s: null pointer", fn);
  const int N = S * S * S;
  const int vec16 = (reinterpret_cast<uintptr_t>(state) & 15) == 0 && game_stride_bytes % 16 == 0;
  int lpg = 1;
  // lanes per game: up to four 16-byte chunks per lane for games of 16 chunks and more (S=9: 16 lanes x 3 chunks, four
  // games per wavefront -- with a wavefront per game 46 lanes did one load each and the launch was latency-bound:
  // 12-16 us for 24 MB), one chunk per lane for the small ones (S=4: 4 lanes)
  while (lpg < 64 && lpg * 16 * (N >= 256 ? 4 : 1) < N) lpg <<= 1;
  const int64_t blocks = (B * lpg + tg::kBlock - 1) / tg::kBlock;
  (void)hipGetLastError();
  hipLaunchKernelGGL(tg::hash_kernel, dim3(grid_for(blocks, 8192)), dim3(tg::kBlock), 0,
                     static_cast<hipStream_t>(stream), state, hash_out, B, N, game_stride_bytes, vec16, lpg);
  return launched(fn);
}

int tg_seen_u64(const uint64_t* keys, uint64_t* table, int64_t capacity, uint8_t* fresh, const uint8_t* mask,
                uint32_t* status, int64_t n, int insert, tg_stream_t stream) {
  const char* fn = "tg_seen_u64";
  if (n < 0) return tg_internal_fail(TG_ERR_INVALID, "%s: n < 0", fn);
  if (capacity < 2 || (capacity & (capacity - 1)) != 0)
    return tg_internal_fail(TG_ERR_INVALID, "%s: capacity=%lld must be a power of two >= 2", fn, (long long)capacity);
  if (!table) return tg_internal_fail(TG_ERR_INVALID, "%s: null table", fn);
  if ((reinterpret_cast<uintptr_t>(table) & 7) || (reinterpret_cast<uintptr_t>(keys) & 7))
    return tg_internal_fail(TG_ERR_INVALID, "%s: keys and table must be 8-byte aligned", fn);
  if (n == 0) return TG_OK;
  if (!keys || (!fresh && !insert)) return tg_internal_fail(TG_ERR_INVALID, "%s: null pointer", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for((n + tg::kBlock - 1) / tg::kBlock, 8192)), block(tg::kBlock);
  (void)hipGetLastError();
  if (fresh)
    hipLaunchKernelGGL(tg::seen_lookup_kernel, grid, block, 0, st, keys, table, static_cast<uint64_t>(capacity - 1), fresh, mask, n);
  if (insert)
    hipLaunchKernelGGL(tg::seen_insert_kernel, grid, block, 0, st, keys, table, static_cast<uint64_t>(capacity - 1), mask, n, status);
  return launched(fn);
}

int tg_rank_i32(const int8_t* state, int32_t* rank_out, int64_t B, int S, int64_t game_stride_bytes,
                tg_stream_t stream) {
  const char* fn = "tg_rank_i32";
  if (int rc = check_state(fn, B, S, game_stride_bytes)) return rc;
  if (B == 0) return TG_OK;
  if (!state || !rank_out) return tg_internal_fail(TG_ERR_INVALID, "%s: null pointer", fn);
  const dim3 grid(grid_for(S == 4 ? (B + 3) / 4 : B, 1 << 20)), block(tg::kBlock);  // (S = 4: a wavefront per game)
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  switch (S) {
    case 4: hipLaunchKernelGGL(tg::rank_kernel<4>, grid, block, 0, st, state, rank_out, B, S, game_stride_bytes); break;
    case 9: hipLaunchKernelGGL(tg::rank_kernel<9>, grid, block, 0, st, state, rank_out, B, S, game_stride_bytes); break;
    case 16: hipLaunchKernelGGL(tg::rank_kernel<16>, grid, block, 0, st, state, rank_out, B, S, game_stride_bytes); break;
    case 25: hipLaunchKernelGGL(tg::rank_kernel<25>, grid, block, 0, st, state, rank_out, B, S, game_stride_bytes); break;
    default: hipLaunchKernelGGL(tg::rank_kernel<0>, grid, block, 0, st, state, rank_out, B, S, game_stride_bytes); break;
  }
  return launched(fn);
}

}  // extern "C"
