// Packed-int16 team kernels (included inside namespace tg by tg_kernels.hip).
//
// The rank-1 update of 16 consecutive state bytes needs, per element, u_i*v_j (constant along a
// row of S elements) times w_l.  Walking an (i,j,l) cursor per element costs ~8 VALU ops and an
// LDS lookup per element.  This form removes both:
//
//  * PERIOD TRICK.  Lane t of a team owns chunks c = t + TSA*n, where TSA is the largest multiple
//    of P = S/gcd(S,16) that fits the team.  16*TSA is a multiple of S, so the position of a
//    lane's 16-element window inside a row, l0 = 16t mod S, is the same for ALL of its chunks:
//    the 16 weights w[(l0+k) mod S] are fetched ONCE per action (8 aligned dword reads from a
//    periodically extended int16 copy of w in LDS) and reused for every chunk of the lane.
//  * A window crosses at most NSEG-1 row boundaries, at lane-constant positions.  The weights are
//    split by lane-constant masks into NSEG vectors, one per row segment; each segment's
//    u_i*v_j is a lane-uniform scalar broadcast into both halves of a dword.
//  * 2 MACs per instruction: acc(int16 x2) = v_pk_mad_i16(uv|uv, w pair, acc).  Bytes <-> int16
//    pairs by v_perm_b32 (sign-extending selectors).
//
// 16-bit sums are exact while nact * f^3 <= 32000 (f = largest |factor|); a prescan of the
// game's tokens decides per workgroup, and games with larger factors run the exact byte-wise
// 32-bit form (slow_game).  The default vocabulary {-1,0,1} always takes the packed path.

// step_many (MODE == MANY) additionally needs, after EVERY step, the zero test and the int8 range
// check.  Checking the range costs more than the MACs, so MANY runs on a lattice instead:
//   x = 256*n + 128 per int16 half, weights pre-multiplied by 256, v_pk_mad_i16 ... clamp.
// An int8 overflow of n is an int16 saturation of x, which knocks x off the lattice (low byte
// != 0x80) for good, because every later increment is a multiple of 256.  So the range check is
// ONE test at the end; a game that fails it is recomputed by the exact byte-wise form before
// anything is stored.  While on the lattice, "state is zero" <=> OR of all x has zero high bytes.
// v_pk_mad_i16 ... clamp forms a*b + c exactly and then saturates, so a single step may move an
// entry by more than 127 as long as it lands in range (-128 + 200 = 72 is exact); the lattice
// only needs representable operands: |u_i v_j| <= 32767 and |256 w_l| <= 32767, i.e. |factor| <= 127.
constexpr uint32_t kLatticeZero = 0x00800080u;

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---- token loading shared by the packed and rows kernels ---------------------------------------
// A game's tokens are `nbytes` contiguous int8 at `src` (any alignment).  They are fetched with
// aligned dword loads: the dword that holds a valid byte never crosses a page, so the <= 3 stray
// bytes at either end are safe to read and are ignored.  All of a lane's loads are independent:
// one memory latency per call.  Returns whether any factor (token - shift) exceeds +-flim.
// With `raw` != nullptr the dwords are also copied to LDS; the first valid byte is raw[head].
template <int TS>
__device__ __forceinline__ int load_tokens_checked(const int8_t* src, int nbytes, int8_t* raw, int lt, int shift,
                                                   int flim, int& head) {
  const uintptr_t A = reinterpret_cast<uintptr_t>(src);
  head = static_cast<int>(A & 3);
  const uint32_t* A4 = reinterpret_cast<const uint32_t*>(A - head);
  const int ndw = (head + nbytes + 3) >> 2;
  uint32_t* rawdw = reinterpret_cast<uint32_t*>(raw);
  int big = 0;
#pragma unroll 4
  for (int idx = lt; idx < ndw; idx += TS) {
    const uint32_t x = A4[idx];
    if (raw) rawdw[idx] = x;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int o = 4 * idx + t - head;
      const int f = sbyte(x, t) - shift;
      big |= (o >= 0 && o < nbytes) && ((f > flim) | (f < -flim));
    }
  }
  return big;
}

// Do all factors of this workgroup's games fit the 16-bit path?  Single tile (nact <= at): the
// check rides on the tile's own raw load (left in LDS, head0 set).  Several tiles: every token is
// scanned first, because the state must not be touched before the decision.  Workgroup-uniform.
// second pass of step_many (ApplyArgs::only_flagged): is game b one of those to redo?  Uniform per game.
template <int MODE>
__device__ __forceinline__ bool flagged_or_all(const ApplyArgs& a, int64_t b) {
  if constexpr (MODE == MANY) {
    // read by every thread before the game's slow_game (which ends with a barrier) rewrites done_step
    return !a.only_flagged || a.done_step[b] == kNeedsExact;
  }
  return true;
}

// Workgroup OR of two flag bits with ONE barrier (HIP's __syncthreads_or costs three and an LDS atomic round trip): every
// wavefront stores its ballots into its own LDS word -- nothing to zero first -- and after the barrier everybody reads
// the kBlock / 64 words.  `slots`: 16 bytes of LDS, 16-byte aligned, not written again before the next barrier.
__device__ __forceinline__ uint32_t block_or2(uint32_t v, uint32_t* slots) {
  static_assert(kBlock == 256, "four wavefront slots");
  const uint32_t w = (__ballot((v & 1u) != 0) ? 1u : 0u) | (__ballot((v & 2u) != 0) ? 2u : 0u);
  if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = w;
  __syncthreads();
  const uint4 q = *reinterpret_cast<const uint4*>(slots);
  return q.x | q.y | q.z | q.w;
}

template <int TS>
__device__ __forceinline__ bool factors_too_large(const int8_t* tok, int nact, int at, int tok_per_action,
                                                  int8_t* raw, int lt, int shift, int flim, int& head0, uint32_t* slots) {
  int big, hd;
  if (nact <= at) {
    big = load_tokens_checked<TS>(tok, nact * tok_per_action, raw, lt, shift, flim, head0);
  } else {
    big = load_tokens_checked<TS>(tok, nact * tok_per_action, nullptr, lt, shift, flim, hd);
  }
  return block_or2(big ? 1u : 0u, slots) != 0;  // (the barrier also completes the raw tokens in LDS)
}

template <int S, int TS>
struct PGeo {
  static constexpr int N = S * S * S;
  static constexpr int G = cgcd(S, 16);
  static constexpr int P = S / G;                       // period of the window position, in chunks
  static constexpr int TSA = (TS / P) * P;              // active lanes per team
  static constexpr int NCHUNK = (N + 15) / 16;
  static constexpr int NCH = (NCHUNK + TSA - 1) / TSA;  // chunks per lane
  static constexpr int NSEG = 1 + (S - G + 15) / S;     // row segments a window can touch
  static constexpr int GPB = kBlock / TS;
  static constexpr int TAIL = N % 16;
  static constexpr int UVLEN = 2 * S + 2;               // shorts: u[S], 0, v[S], pad
  static constexpr int WE = (S + 16 + 1) & ~1;          // shorts per periodic copy of w (even)
  static constexpr int FSTRIDE = UVLEN + 2 * WE;        // shorts per action (even: dword aligned)
  static constexpr int ATILE_RAW = 32768 / (GPB * FSTRIDE * 2);
  static constexpr int ATILE = ATILE_RAW > 64 ? 64 : ATILE_RAW;
  static constexpr int LDS_SHORTS = GPB * ATILE * FSTRIDE;
  static_assert(TSA >= 1 && NSEG >= 1 && NSEG <= 4 && ATILE >= 1, "geometry");
};

__device__ __forceinline__ uint32_t pk_mad_i16(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// saturating form: the int16 result clamps to [-32768, 32767]
__device__ __forceinline__ uint32_t pk_mad_i16_sat(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// the same with the LOW / HIGH half of `a` used for both halves of the product (VOP3P op_sel)
__device__ __forceinline__ uint32_t pk_mad_i16_sat_lo(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t pk_mad_i16_sat_hi(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t pk_lshl8_b16(uint32_t a) {
  uint32_t d;
  // the shift count is per half: an inline constant 8 would shift the low half only
  asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(d) : "v"(0x00080008u), "v"(a));
  return d;
}
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t pk_mul_lo_u16(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// 16 int8 (one uint4) -> 8 dwords of two sign-extended int16 each, element order preserved
__device__ __forceinline__ void unpack_pairs(const uint4& q, uint32_t (&A)[8]) {
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t x = w[d], y = x << 8;  // y.byte1 = x.byte0, y.byte3 = x.byte2
    A[2 * d] = __builtin_amdgcn_perm(x, y, 0x0A050804u);      // b0, sign(b0), b1, sign(b1)
    A[2 * d + 1] = __builtin_amdgcn_perm(x, y, 0x0B070906u);  // b2, sign(b2), b3, sign(b3)
  }
}

// low bytes of the 16 int16 -> uint4; nz |= packed bytes; ovf |= (x+128) per half (bits 8..15 set when out of range)
__device__ __forceinline__ uint4 pack_pairs(const uint32_t (&A)[8], uint32_t& nz, uint32_t& ovf) {
  uint32_t w[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    ovf |= pk_add_u16(A[2 * d], 0x00800080u) | pk_add_u16(A[2 * d + 1], 0x00800080u);
    w[d] = __builtin_amdgcn_perm(A[2 * d + 1], A[2 * d], 0x06040200u);
    nz |= w[d];
  }
  return uint4{w[0], w[1], w[2], w[3]};
}

// Dynamic LDS (bytes) for `at` staged actions per tile: int16 tables, raw token bytes, flags.
// KEYS with a workgroup per parent (TS = 256): + one 64-bit partial key sum per child of the tile and wavefront.
template <int S, int TS, int MODE, bool KEYS = false>
constexpr int packed_lds_bytes(int at) {
  using G = PGeo<S, TS>;
  const int tables = G::GPB * at * G::FSTRIDE * 2;
  const int raw = G::GPB * ((at * 3 * S + 8 + 3) & ~3);
  const int nflag = MODE == MANY ? TG_MAX_ACTIONS + 16 : cmax(4, 3 * G::GPB * at);  // MANY: + recompute byte per team
  const int keysum = (KEYS && TS == kBlock) ? at * 32 : 0;
  return ((tables + raw + ((nflag + 3) & ~3) + 15) & ~15) + keysum + 32;  // + two 16-byte slot arrays of block_or2
}

// NTS (EXPAND at S = 16 only): the children leave by non-temporal stores (268 MB of children: 56.5 -> 50.6 us; the
// 15 625-byte children of S = 25 end in partial lines and lose with them: 124 -> 134 us).
// KEYS (round 4; EXPAND at S = 16 and S = 25, tg_expand_keyed_i8): the 64-bit key of every child is formed while the child is
// in registers -- extend_tree filters every expansion against the tree (act.py:183-195), and a second pass re-reads the
// 268 MB (S = 16, 8 192 parents x 8) / 512 MB (S = 25, 4 096 x 8) of children it has just written.  S = 16: a team is a
// wavefront, the chunk sums meet by shuffles.  S = 25: a team is the workgroup -- every wavefront leaves its partial sum per
// child in LDS and, behind the barrier the child loop ends with anyway, thread k adds the four of child k.
template <int S, int TS, int MODE, bool NTS = false, bool KEYS = false>
__global__ __launch_bounds__(kBlock) void packed_kernel(ApplyArgs a, int flim, int at) {
  using G = PGeo<S, TS>;
  static_assert(!KEYS || (MODE == EXPAND && (TS == 64 || TS == kBlock)), "keys: a wavefront or a workgroup per parent");
  constexpr bool SUB = (MODE != GENF);  // STEP, MANY, EXPAND subtract
  extern __shared__ __attribute__((aligned(16))) short lds[];
  const int raw_stride = (at * 3 * S + 8 + 3) & ~3;  // bytes of raw tokens per team
  int8_t* const raw_all = reinterpret_cast<int8_t*>(lds + G::GPB * at * G::FSTRIDE);
  uint8_t* const flags = reinterpret_cast<uint8_t*>(raw_all + G::GPB * raw_stride);
  // the last 32 bytes of the dynamic LDS (16-byte aligned): two slot arrays for block_or2
  uint32_t* const or_slots = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(lds) + packed_lds_bytes<S, TS, MODE, KEYS>(at) - 32);
  // (KEYS, TS = 256) hpart[k * 4 + wave]: the wavefront's share of child k's key sum; in front of the OR slots, 16-byte aligned
  unsigned long long* const hpart = reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(or_slots) - at * 32);
  (void)hpart;

  const int tid = threadIdx.x;
  const int team = tid / TS, lt = tid % TS;
  int64_t g = static_cast<int64_t>(blockIdx.x) * G::GPB + team;
  bool live = g < a.B;
  if (!live) g = a.B - 1;
  if constexpr (MODE == MANY) {
    if (a.only_flagged) {  // second pass: only the games many_mfma_kernel handed over
      live = live && a.done_step[g] == kNeedsExact;
      if (!__syncthreads_or(live)) return;
    }
  }
  const int8_t* const tok = a.actions + g * a.nact * (3 * S);
  int8_t* const raw = raw_all + team * raw_stride;

  // Geometry and the state loads come FIRST: the loads do not depend on the tokens, so their
  // latency overlaps the token fetch and the fallback decision below instead of following it.
  // ---- lane geometry ----------------------------------------------------------------------------
  const bool active = lt < G::TSA;
  const int l0 = (16 * lt) % S;
  const int woff = G::UVLEN + (l0 & 1) * G::WE + (l0 & ~1);  // window start (shorts) inside an action's table
  uint32_t mask[G::NSEG > 1 ? G::NSEG - 1 : 1][8];
#pragma unroll
  for (int s = 0; s + 1 < G::NSEG; ++s) {
    const int lo = (s == 0) ? -1000 : s * S - l0, hi = (s + 1) * S - l0;  // segment s covers k in [lo,hi)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int k0 = 2 * p, k1 = 2 * p + 1;
      mask[s][p] = ((k0 >= lo && k0 < hi) ? 0x0000FFFFu : 0u) | ((k1 >= lo && k1 < hi) ? 0xFFFF0000u : 0u);
    }
  }
  bool cv[G::NCH], ctail[G::NCH];
  int rowidx[G::NCH][G::NSEG];  // (index of u_i) | (index of v_j) << 16, both into the action's table
#pragma unroll
  for (int n = 0; n < G::NCH; ++n) {
    const int c = lt + G::TSA * n;
    cv[n] = active && c < G::NCHUNK;
    ctail[n] = (G::TAIL != 0) && (c == G::NCHUNK - 1);
    const int r0 = (16 * c) / S;
#pragma unroll
    for (int s = 0; s < G::NSEG; ++s) {
      const int row = r0 + s;
      int i = row / S;
      const int j = row - i * S;
      if (!cv[n] || i >= S) i = S;  // table[S] == 0: rows past the tensor (and idle chunks) add nothing
      rowidx[n][s] = i | ((S + 1 + j) << 16);
    }
  }

  // ---- state ------------------------------------------------------------------------------------
  uint4 par[G::NCH];
#pragma unroll
  for (int n = 0; n < G::NCH; ++n) {
    par[n] = uint4{0, 0, 0, 0};
    if (MODE != GENF && cv[n])
      par[n] = load_chunk<G::TAIL>(a.in + g * a.in_stride + 16 * (lt + G::TSA * n), ctail[n]);
  }

  auto load_raw = [&](int a0, int na, int& head) {  // tokens of actions [a0,a0+na) -> LDS, unchecked
    (void)load_tokens_checked<TS>(tok + a0 * (3 * S), na * 3 * S, raw, lt, a.shift, 0x7fffffff, head);
  };

  int head0 = 0;
  if (factors_too_large<TS>(tok, a.nact, at, 3 * S, raw, lt, a.shift, flim, head0, or_slots)) {
    // exact byte-wise form, one game at a time (rare; speed is irrelevant)
    note_fallback();
    for (int t = 0; t < G::GPB; ++t) {
      const int64_t b = static_cast<int64_t>(blockIdx.x) * G::GPB + t;
      if (b < a.B && flagged_or_all<MODE>(a, b)) slow_game<MODE>(a, b, flags);
    }
    if constexpr (KEYS) {  // the exact form wrote the children to memory: every team keys those of its parent from there
      __syncthreads();
      for (int k = 0; k < a.nact; ++k) {  // (workgroup-uniform trip count: the TS = 256 form has barriers inside)
        const int64_t child = g * a.nact + k;
        uint64_t hk = 0;
        if (live) {
#pragma unroll
          for (int n = 0; n < G::NCH; ++n)
            if (cv[n]) hk += hash_chunk(load_chunk<G::TAIL>(a.out + child * a.out_stride + 16 * (lt + G::TSA * n), ctail[n]), lt + G::TSA * n);
        }
        for (int off = 32; off > 0; off >>= 1) hk += __shfl_xor(hk, off);
        if constexpr (TS == 64) {
          if (lt == 0 && live) a.keys[child] = hash_finish(hk, S * S * S);
        } else {
          if ((tid & 63) == 0) hpart[tid >> 6] = hk;
          __syncthreads();
          if (tid == 0 && live) a.keys[child] = hash_finish(hpart[0] + hpart[1] + hpart[2] + hpart[3], S * S * S);
          __syncthreads();
        }
      }
    }
    return;
  }

  short* const F = lds + team * (at * G::FSTRIDE);
  // Which token feeds table entry `pos` is the same for every action, and a lane always writes
  // the same entries (pos = lt + e*TS): resolve the mapping once.  src < 0: the entry is a zero.
  constexpr int EPL = (G::FSTRIDE + TS - 1) / TS;  // table entries per lane per action
  int src[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int pos = lt + e * TS;
    int sidx = -1;
    if (pos < S) {
      sidx = pos;                                  // u (stored negated for the subtracting modes)
    } else if (pos > S && pos <= 2 * S) {
      sidx = S + (pos - S - 1);                    // v
    } else if (pos >= G::UVLEN && pos < G::FSTRIDE) {
      int q = pos - G::UVLEN;
      const int copy = q >= G::WE;
      q -= copy * G::WE;
      sidx = 2 * S + (q + copy) % S;               // periodic copies of w (copy 1 shifted by one)
    }
    src[e] = sidx;
  }
  const bool negate_u = SUB;
  // tile [a0,a0+na): raw tokens (already in LDS when `loaded`) -> int16 tables u,0,v,pad, w-periodic x2
  auto stage = [&](int a0, int na, bool loaded) {
    int head = head0;
    if (!loaded) {
      __syncthreads();  // previous tile's tables and raw bytes are no longer read
      load_raw(a0, na, head);
      __syncthreads();
    }  // (loaded: the barrier inside factors_too_large has already completed the raw tokens)
    for (int k = 0; k < na; ++k) {
      const int8_t* t = raw + head + k * (3 * S);
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int pos = lt + e * TS;
        if (pos < G::FSTRIDE) {
          int val = 0;
          if (src[e] >= 0) {
            val = t[src[e]] - a.shift;
            if (negate_u && pos < S) val = -val;
            if (MODE == MANY && pos >= G::UVLEN) val *= 256;  // lattice weights
          }
          F[k * G::FSTRIDE + pos] = static_cast<short>(val);
        }
      }
    }
    __syncthreads();
  };

  // the lane's 16 weights for one action, split into the NSEG row segments of its window
  auto window = [&](const short* Fa, uint32_t (&ws)[G::NSEG][8]) {
    const uint32_t* wp = reinterpret_cast<const uint32_t*>(Fa + woff);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const uint32_t w2 = wp[p];
      if constexpr (G::NSEG == 1) {
        ws[0][p] = w2;
      } else if constexpr (G::NSEG == 2) {
        ws[0][p] = w2 & mask[0][p];
        ws[1][p] = w2 & ~mask[0][p];
      } else {
        uint32_t rest = w2;
#pragma unroll
        for (int s = 0; s + 1 < G::NSEG; ++s) {
          ws[s][p] = w2 & mask[s][p];
          rest ^= ws[s][p];
        }
        ws[G::NSEG - 1][p] = rest;
      }
    }
  };

  auto apply = [&](const short* Fa, uint32_t (&A)[G::NCH][8]) {  // A += rank-1 term of one action
    uint32_t ws[G::NSEG][8];
    window(Fa, ws);
#pragma unroll
    for (int n = 0; n < G::NCH; ++n) {
#pragma unroll
      for (int s = 0; s < G::NSEG; ++s) {
        const int uv = mul24_pinned(Fa[rowidx[n][s] & 0xffff], Fa[rowidx[n][s] >> 16]);
        const uint32_t pr = __builtin_amdgcn_perm(static_cast<uint32_t>(uv), static_cast<uint32_t>(uv), 0x05040100u);
#pragma unroll
        for (int p = 0; p < 8; ++p)
          A[n][p] = (MODE == MANY) ? pk_mad_i16_sat(pr, ws[s][p], A[n][p]) : pk_mad_i16(pr, ws[s][p], A[n][p]);
      }
    }
  };

  uint32_t ovf = 0;

  if constexpr (MODE == STEP) {
    stage(0, 1, true);
    uint32_t A[G::NCH][8];
#pragma unroll
    for (int n = 0; n < G::NCH; ++n) unpack_pairs(par[n], A[n]);
    apply(F, A);
    uint32_t nz = 0;
    const bool inplace = a.in == a.out;
#pragma unroll
    for (int n = 0; n < G::NCH; ++n) {
      const uint4 q = pack_pairs(A[n], nz, ovf);
      // in place, a chunk the action did not touch (u_i v_j = 0 on its rows: most chunks under the
      // reference's factor distribution) needs no store at all
      const bool same = inplace && q.x == par[n].x && q.y == par[n].y && q.z == par[n].z && q.w == par[n].w;
      if (live && cv[n] && !same)
        store_chunk<G::TAIL>(a.out + g * a.out_stride + 16 * (lt + G::TSA * n), q, ctail[n]);
    }
    bool any_nz, any_ovf;
    if constexpr (TS == 256) {
      const uint32_t both = block_or2((nz != 0 ? 1u : 0u) | ((ovf & 0xFF00FF00u) != 0 ? 2u : 0u), or_slots + 4);
      any_nz = (both & 1u) != 0;
      any_ovf = (both & 2u) != 0;
    } else {
      any_nz = team_any<TS>(nz != 0);
      any_ovf = team_any<TS>((ovf & 0xFF00FF00u) != 0);
    }
    if (lt == 0 && live) {
      a.done[g] = any_nz ? 0 : 1;
      if (a.overflow && any_ovf) a.overflow[g] = 1;
    }
  } else if constexpr (MODE == MANY || MODE == GENF) {
    uint32_t A[G::NCH][8];
#pragma unroll
    for (int n = 0; n < G::NCH; ++n) {
      unpack_pairs(par[n], A[n]);
      if constexpr (MODE == MANY) {
#pragma unroll
        for (int p = 0; p < 8; ++p) A[n][p] = pk_add_u16(pk_lshl8_b16(A[n][p]), kLatticeZero);
      }
    }
    int done_step = -1;
    if constexpr (MODE == MANY && TS == 256) {
      for (int k = tid; k < a.nact; k += kBlock) flags[k] = 0;
    }
    for (int a0 = 0; a0 < a.nact; a0 += at) {
      const int na = min(at, a.nact - a0);
      stage(a0, na, a.nact <= at);
      for (int k = 0; k < na; ++k) {
        apply(F + k * G::FSTRIDE, A);
        if constexpr (MODE == MANY) {
          uint32_t nz = 0;
#pragma unroll
          for (int n = 0; n < G::NCH; ++n)
#pragma unroll
            for (int p = 0; p < 8; ++p) nz |= A[n][p];
          if constexpr (TS == 256) {
            if (nz & 0xFF00FF00u) flags[a0 + k] = 1;
          } else {
            if (!team_any<TS>((nz & 0xFF00FF00u) != 0) && done_step < 0) done_step = a0 + k;
          }
        }
      }
    }
    if constexpr (MODE == MANY) {
      // off the lattice <=> some step overflowed int8: recompute those games exactly, store nothing
      uint32_t off = 0;
#pragma unroll
      for (int n = 0; n < G::NCH; ++n)
#pragma unroll
        for (int p = 0; p < 8; ++p) off |= (A[n][p] ^ kLatticeZero) & 0x00FF00FFu;
      bool bad;
      if constexpr (TS == 256) {
        bad = __syncthreads_or(off != 0);
        if (bad) {
          note_fallback();
          if (live) slow_game<MODE>(a, g, flags);  // live is workgroup-uniform here (one game per workgroup)
          return;
        }
      } else {
        bad = team_any<TS>(off != 0);
        uint8_t* const badF = flags + TG_MAX_ACTIONS;  // one byte per team, past slow_game's per-step flags
        __syncthreads();
        if (lt == 0) badF[team] = bad && live;
        __syncthreads();
        for (int t = 0; t < G::GPB; ++t)
          if (badF[t]) {
            note_fallback();
            slow_game<MODE>(a, static_cast<int64_t>(blockIdx.x) * G::GPB + t, flags);
          }
        if (bad) return;  // after the loop: slow_game needs every lane of the workgroup
      }
#pragma unroll
      for (int n = 0; n < G::NCH; ++n) {
        uint4 q;
        uint32_t w[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) w[d] = __builtin_amdgcn_perm(A[n][2 * d + 1], A[n][2 * d], 0x07050301u);
        q = uint4{w[0], w[1], w[2], w[3]};
        if (live && cv[n]) store_chunk<G::TAIL>(a.out + g * a.out_stride + 16 * (lt + G::TSA * n), q, ctail[n]);
      }
      if constexpr (TS == 256) {
        __syncthreads();  // orders the per-step flag writes
        if (tid == 0) {
          for (int k = 0; k < a.nact; ++k)
            if (!flags[k]) { done_step = k; break; }
        }
      }
      if (lt == 0 && live) a.done_step[g] = done_step;
    } else {
      uint32_t nz = 0;
#pragma unroll
      for (int n = 0; n < G::NCH; ++n) {
        const uint4 q = pack_pairs(A[n], nz, ovf);  // GENF: the only range check (sum narrowed once)
        if (live && cv[n]) store_chunk<G::TAIL>(a.out + g * a.out_stride + 16 * (lt + G::TSA * n), q, ctail[n]);
      }
      bool any_ovf;
      if constexpr (TS == 256) {
        any_ovf = __syncthreads_or((ovf & 0xFF00FF00u) != 0);
      } else {
        any_ovf = team_any<TS>((ovf & 0xFF00FF00u) != 0);
      }
      if (lt == 0 && live && a.overflow && any_ovf) a.overflow[g] = 1;
    }
  } else {  // EXPAND
    uint8_t* const nzF = flags + team * (3 * at);
    uint8_t* const ovF = nzF + at;
    uint8_t* const nnF = ovF + at;
    for (int a0 = 0; a0 < a.nact; a0 += at) {
      const int na = min(at, a.nact - a0);
      stage(a0, na, a.nact <= at);
      for (int k = lt; k < na; k += TS) {  // null action <=> u, v or w is the zero vector
        const short* Fa = F + k * G::FSTRIDE;
        int nu = 0, nv = 0, nw = 0;
        for (int s = 0; s < S; ++s) {
          nu |= Fa[s];
          nv |= Fa[S + 1 + s];
          nw |= Fa[G::UVLEN + s];
        }
        nnF[k] = (nu != 0) && (nv != 0) && (nw != 0);
        nzF[k] = 0;
        ovF[k] = 0;
      }
      __syncthreads();
      for (int k = 0; k < na; ++k) {
        const int64_t child = g * a.nact + a0 + k;
        uint32_t A[G::NCH][8];
#pragma unroll
        for (int n = 0; n < G::NCH; ++n) unpack_pairs(par[n], A[n]);
        apply(F + k * G::FSTRIDE, A);
        uint32_t nz = 0, covf = 0;
        uint64_t hk = 0;
#pragma unroll
        for (int n = 0; n < G::NCH; ++n) {
          const uint4 q = pack_pairs(A[n], nz, covf);
          if (live && cv[n])
          {
            int8_t* const dp = a.out + child * a.out_stride + 16 * (lt + G::TSA * n);
            if constexpr (NTS && G::TAIL == 0) store16_nt(dp, q);
            else store_chunk<G::TAIL>(dp, q, ctail[n]);
          }
          if constexpr (KEYS) {
            if (cv[n]) hk += hash_chunk(q, lt + G::TSA * n);
          }
        }
        if constexpr (KEYS) {
          for (int off = 32; off > 0; off >>= 1) hk += __shfl_xor(hk, off);
          if constexpr (TS == 64) {
            if (lt == 0 && live) a.keys[child] = hash_finish(hk, S * S * S);
          } else {
            if ((tid & 63) == 0) hpart[4 * k + (tid >> 6)] = hk;  // met behind the tile's closing barrier
          }
        }
        if (nz) nzF[k] = 1;
        if (covf & 0xFF00FF00u) ovF[k] = 1;
      }
      __syncthreads();
      for (int k = lt; k < na; k += TS) {
        if (!live) continue;
        const int64_t child = g * a.nact + a0 + k;
        if constexpr (KEYS && TS == kBlock)
          a.keys[child] = hash_finish(hpart[4 * k] + hpart[4 * k + 1] + hpart[4 * k + 2] + hpart[4 * k + 3], S * S * S);
        a.done[child] = nzF[k] ? 0 : 1;
        if (a.changed) a.changed[child] = nnF[k];
        if (a.overflow && ovF[k]) a.overflow[child] = 1;
      }
    }
  }
}

// number of non-zero bytes of a dword, accumulated: TWO instructions.  v_msad_u8 sums |a_i - b_i| over the bytes with
// b_i != 0, and x ^ 0x01010101 differs from x by exactly one in every byte (four with the mask-and-popcount form)
__device__ __forceinline__ int nz_bytes(uint32_t x, int acc) {
  return static_cast<int>(__builtin_amdgcn_msad_u8(x ^ 0x01010101u, x, static_cast<uint32_t>(acc)));
}
__device__ __forceinline__ int nz_bytes16(const uint4& q) { return nz_bytes(q.x, nz_bytes(q.y, nz_bytes(q.z, nz_bytes(q.w, 0)))); }

// =============================================================================================
// S = 25 single step without the staging round trip of packed_kernel (STEP, one action, one workgroup per game).
// packed_kernel brings the raw tokens to LDS, scans them (barrier), turns them into the int16 tables (barrier),
// computes, and ORs done / overflow (barrier).  Here each of the 136 table entries is loaded straight from the
// game's 75 tokens by the thread that writes it -- issued together with the thread's four state chunks, so the
// wavefront's chain is ONE memory round trip -> table -> barrier -> arithmetic -> stores -> the done / overflow OR.
// No range prescan: the saturating 16-bit form (as in s16_step_kernel) is exact or lands outside int8, and a chunk
// whose range test fails is redone by its lane in 32 bits from the same tables (|factor| <= 255: host-checked shift).
// Geometry and tables are packed_kernel's (period trick, two row segments per window).
// Arithmetic only where the action acts, as in s16_step_kernel: a chunk is a candidate when one of the (at most two)
// rows it touches has u_i v_j != 0 (~14 % of the chunks under the reference's factor distribution); each wavefront
// compacts its candidates (16 bytes, chunk index, the two products) into a 64-entry LDS queue, and in one dense pass
// lane k takes entry k: window from the table at the chunk's own position, arithmetic, store.  Unchanged chunks only
// feed the zero test.  A wavefront with more than 64 candidates does its chunks directly.
// =============================================================================================
//
// LINES / NTL (batches that stream from HBM, as s16_step_kernel's): the dense pass hands its results back through the
// queue and every lane stores its own chunks at 128-byte-LINE granularity -- a chunk is stored when any chunk of its
// line changed (a game starts o = (address / 16) mod 8 chunks into a line -- o = g mod 8 on the 15 632-byte stride, 0 on a
// 128-byte-multiple stride -- so the eight chunks of a line are eight consecutive lanes from lane (-(o + 250 n)) mod 8 on;
// lines that straddle two wavefronts or two games stay partial) -- and, beyond the Infinity Cache, the state is read by
// non-temporal loads.
// TRACK (tg_step_tracked_i8 below the batch size where its sparse kernel pays): the same full step with the game's carried
// count of non-zero entries (a.done_step doubles as the nnz array) updated from the candidate chunks, done = (nnz == 0).
template <bool LINES, bool NTL, bool TRACK = false>
__global__ __launch_bounds__(kBlock, LINES ? 6 : 7) void s25_step_kernel(ApplyArgs a) {  // (6 spills at 64 VGPRs)
  static_assert(!TRACK || !LINES, "the tracked form is the cache-resident variant");
  constexpr int S = 25;
  using G = PGeo<S, kBlock>;
  static_assert(G::NSEG == 2 && G::NCH == 4 && G::FSTRIDE <= kBlock, "s25_step_kernel geometry");
  constexpr int QCAP = 64;
  __shared__ __attribute__((aligned(16))) short F[(G::FSTRIDE + 7) & ~7];
  __shared__ __attribute__((aligned(16))) uint32_t or_slots[4];
  __shared__ __attribute__((aligned(16))) uint4 qd[kBlock / 64][QCAP];  // candidate chunks
  __shared__ __attribute__((aligned(16))) int4 qm[kBlock / 64][QCAP];   // (chunk index, uv of its first row, of its second row, -)
  __shared__ __attribute__((aligned(8))) int uvt[S * S + 3];            // -u_i v_j per row (i, j), 0 behind the last
  __shared__ int dsum;                                                  // TRACK: the workgroup's nnz delta
  const int lt = threadIdx.x, lane = lt & 63, wave = lt >> 6;
  const int64_t g = sweep_index(blockIdx.x, gridDim.x, a.sweep);
  const int nnz_in = (TRACK && lt == 0) ? a.done_step[g] : 0;
  if (TRACK && lt == 0) dsum = 0;  // (the barrier behind the tables orders it)
  int delta = 0;
  (void)delta;
  const int8_t* const tok = a.actions + g * (3 * S);
  const int8_t* const in = a.in + g * a.in_stride;
  int8_t* const out = a.out + g * a.out_stride;
  // ---- every load of the thread is issued before anything is used: its token FIRST (vmcnt retires in order: the
  // table and the barrier then run while the four state chunks are still on their way) ----
  // table entry `lt`: -u[0..S), 0, v[0..S), pad, two periodic copies of w (the second shifted by one)
  int sidx = -1;
  if (lt < S) {
    sidx = lt;
  } else if (lt > S && lt <= 2 * S) {
    sidx = S + (lt - S - 1);
  } else if (lt >= G::UVLEN && lt < G::FSTRIDE) {
    int q = lt - G::UVLEN;
    const int copy = q >= G::WE;
    q -= copy * G::WE;
    sidx = 2 * S + (q + copy) % S;
  }
  const int tokv = tok[sidx >= 0 ? sidx : 0];
  // round 3: the product -u_i v_j of every ROW (i, j), once per game instead of twice per chunk: thread lt takes rows
  // lt, lt + 256, lt + 512 (< S^2) straight from the tokens (two byte loads per row, issued here with everything else),
  // and a chunk's classification below is one division and two LDS words instead of two divisions, two modulos, four
  // table reads and two multiplies (the step is bound by instruction issue and barrier latency, not by bandwidth: at
  // 2 GiB of states 3 to 7 resident workgroups per CU and every store form land on the same 580-610 us)
  constexpr int NROWT = (S * S + kBlock - 1) / kBlock;
  int tu[NROWT], tv[NROWT];
#pragma unroll
  for (int k = 0; k < NROWT; ++k) {
    const int row = lt + kBlock * k, rr = row < S * S ? row : S * S - 1;
    const int i = rr / S, j = rr - i * S;
    tu[k] = tok[i];
    tv[k] = tok[S + j];
  }
  const bool active = lt < G::TSA;
  // (four named chunks, not an array: see s16_step_kernel)
  auto load = [&](int n) {
    const int c = lt + G::TSA * n;
    uint4 q = uint4{0, 0, 0, 0};
    if (active && c < G::NCHUNK) {
      if (NTL && !(G::TAIL != 0 && c == G::NCHUNK - 1)) {
        const v4u_t v = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(in + 16 * c));
        q = uint4{v.x, v.y, v.z, v.w};
      } else {
        q = load_chunk<G::TAIL>(in + 16 * c, G::TAIL != 0 && c == G::NCHUNK - 1);
      }
    }
    return q;
  };
  const uint4 p0 = load(0), p1 = load(1), p2 = load(2), p3 = load(3);
  if (lt < G::FSTRIDE) {
    int val = sidx >= 0 ? tokv - a.shift : 0;
    if (lt < S) val = -val;
    F[lt] = static_cast<short>(val);
  }
#pragma unroll
  for (int k = 0; k < NROWT; ++k) {
    const int row = lt + kBlock * k;
    if (row < S * S) uvt[row] = mul24_pinned(a.shift - tu[k], tv[k] - a.shift);  // -u_i v_j
  }
  if (lt == 0) uvt[S * S] = 0;  // the row behind the last one (the last chunk's second row)
  const bool inplace = a.in == a.out;
  uint32_t nz = 0, ovf = 0;
  auto differs = [](const uint4& x, const uint4& y) { return x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w; };
  // one chunk c with the products of its two rows: x + uv * w over the window at the chunk's position
  auto chunk = [&](const uint4& x, int c, int uv0, int uv1, uint32_t& cnz) {
    const int l0 = (16 * c) % S, hi = S - l0;  // elements k < hi belong to the window's first row
    const uint32_t* wp = reinterpret_cast<const uint32_t*>(F + G::UVLEN + (l0 & 1) * G::WE + (l0 & ~1));
    uint32_t wraw[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) wraw[p] = wp[p];
    // |u v| may exceed int16: clamp the multiplier; the product then saturates and the range test below fails
    const int c0 = max(-32767, min(32767, uv0)), c1 = max(-32767, min(32767, uv1));
    const uint32_t pr0 = __builtin_amdgcn_perm(static_cast<uint32_t>(c0), static_cast<uint32_t>(c0), 0x05040100u);
    const uint32_t pr1 = __builtin_amdgcn_perm(static_cast<uint32_t>(c1), static_cast<uint32_t>(c1), 0x05040100u);
    uint32_t A[8];
    unpack_pairs(x, A);
#pragma unroll
    for (int p = 0; p < 8; ++p) {  // the pair's multiplier: first row below hi, second row from hi on (one pair may straddle)
      const uint32_t pr = (2 * p + 1 < hi) ? pr0 : ((2 * p >= hi) ? pr1 : __builtin_amdgcn_perm(pr1, pr0, 0x07060100u));
      A[p] = pk_mad_i16_sat(pr, wraw[p], A[p]);
    }
    uint32_t c16 = 0;
    cnz = 0;
    uint4 res = pack_pairs(A, cnz, c16);
    if (__builtin_expect((c16 & 0xFF00FF00u) != 0, 0)) {  // rare: the chunk again in 32 bits (wrapped bytes + flag)
      const uint32_t pd[4] = {x.x, x.y, x.z, x.w};
      uint32_t rd[4];
      int o32 = 0;
      cnz = 0;
#pragma unroll
      for (int d = 0; d < 4; ++d) {  // (unrolled: a dynamically indexed wraw[] would be moved to LDS)
        int e[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int k = 4 * d + t;
          const int w = static_cast<short>(wraw[k >> 1] >> (16 * (k & 1)));
          e[t] = sbyte(pd[d], t) + (k < hi ? uv0 : uv1) * w;
          o32 |= e[t] + 128;
        }
        rd[d] = pack4(e[0], e[1], e[2], e[3]);
        cnz |= rd[d];
      }
      res = uint4{rd[0], rd[1], rd[2], rd[3]};
      ovf |= static_cast<uint32_t>(o32) & ~255u;
    }
    return res;
  };
  __syncthreads();

  // ---- which of the lane's chunks does the action touch?  candidates -> the wavefront's queue ----
  int total = 0;  // wave-uniform
  auto enqueue = [&](int n, const uint4& pn, int& uv0, int& uv1, int& slot) {
    const int c = lt + G::TSA * n;
    const bool cv = active && c < G::NCHUNK;
    const int cc = cv ? c : 0;                        // (idle chunks read row 0 and are zeroed below)
    const int r0 = (16 * cc) / S, l0 = 16 * cc - S * r0;  // row (i, j) of the chunk's first element; the second row is r0 + 1
    uv0 = cv ? uvt[r0] : 0;
    uv1 = (cv && l0 + 16 > S) ? uvt[r0 + 1] : 0;      // 0 when the window does not reach the second row
    const bool cand = (uv0 | uv1) != 0;
    const unsigned long long m = __ballot(cand);
    slot = total + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                              __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)));
    if (cand) {
      if (slot < QCAP) {
        qd[wave][slot] = pn;
        qm[wave][slot] = int4{c, uv0, uv1, 0};
      }
    } else {
      nz |= pn.x | pn.y | pn.z | pn.w;
      if (!LINES && !inplace && cv) store_chunk<G::TAIL>(out + 16 * c, pn, G::TAIL != 0 && c == G::NCHUNK - 1);
    }
    total += __builtin_popcountll(m);
  };
  int u00, u01, u10, u11, u20, u21, u30, u31, s0, s1, s2, s3;
  enqueue(0, p0, u00, u01, s0);
  enqueue(1, p1, u10, u11, s1);
  enqueue(2, p2, u20, u21, s2);
  enqueue(3, p3, u30, u31, s3);
  auto finish = [&](const uint4& x, int c, int uv0, int uv1) {
    uint32_t cnz;
    const uint4 res = chunk(x, c, uv0, uv1, cnz);
    nz |= cnz;
    if constexpr (TRACK) delta += nz_bytes16(res) - nz_bytes16(x);  // (the tail chunk's bytes past the tensor are zero in both)
    // in place, a chunk the action left as it was needs no store
    if (!inplace || differs(res, x)) store_chunk<G::TAIL>(out + 16 * c, res, G::TAIL != 0 && c == G::NCHUNK - 1);
  };
  if (total <= QCAP) {
    // ---- dense pass: lane k takes entry k (LDS serves one wavefront's accesses in order: no barrier) ----
    __builtin_amdgcn_wave_barrier();
    if (lane < total) {
      const uint4 x = qd[wave][lane];
      const int4 me = qm[wave][lane];
      if constexpr (LINES) {
        uint32_t cnz;
        qd[wave][lane] = chunk(x, me.x, me.y, me.z, cnz);  // back to the owner, who stores whole lines
        nz |= cnz;
      } else {
        finish(x, me.x, me.y, me.z);
      }
    }
    if constexpr (LINES) {
      __builtin_amdgcn_wave_barrier();
      auto own = [&](int n, const uint4& pn, bool cand, int slot) {
        const int c = lt + G::TSA * n;
        const bool cv = active && c < G::NCHUNK;
        uint4 res = pn;
        if (cand) res = qd[wave][slot];
        const bool chg = cv && differs(res, pn);
        bool st = cv;  // out of place everything is written
        if (inplace) {
          // the lanes of my line: (g + c) >> 3 equal, i.e. eight consecutive lanes from a multiple of 8 minus sh on
          const unsigned long long m = __ballot(chg);
          const int sh = static_cast<int>(((reinterpret_cast<uintptr_t>(out) >> 4) + G::TSA * n) & 7);  // any 16-byte-multiple stride
          const int first = ((lane + sh) & ~7) - sh;  // may be negative: the line began in the previous wavefront
          const int lo = max(first, 0), hi = min(first + 8, 64);
          st = ((m >> lo) & ((1ull << (hi - lo)) - 1ull)) != 0;
        }
        if (cv && st) store_chunk<G::TAIL>(out + 16 * c, res, G::TAIL != 0 && c == G::NCHUNK - 1);  // (cv: idle lanes share lines too)
      };
      own(0, p0, (u00 | u01) != 0, s0);
      own(1, p1, (u10 | u11) != 0, s1);
      own(2, p2, (u20 | u21) != 0, s2);
      own(3, p3, (u30 | u31) != 0, s3);
    }
  } else {
    // ---- dense factors: every candidate chunk by its own lane (16-byte stores in every variant) ----
    auto direct = [&](int n, const uint4& pn, int uv0, int uv1) {
      const int c = lt + G::TSA * n;
      if ((uv0 | uv1) != 0) finish(pn, c, uv0, uv1);
      else if (LINES && !inplace && active && c < G::NCHUNK) store_chunk<G::TAIL>(out + 16 * c, pn, G::TAIL != 0 && c == G::NCHUNK - 1);
    };
    direct(0, p0, u00, u01);
    direct(1, p1, u10, u11);
    direct(2, p2, u20, u21);
    direct(3, p3, u30, u31);
  }
  if constexpr (TRACK) {
    delta = wave_sum(delta);
    if (lane == 0 && delta) atomicAdd(&dsum, delta);
  }
  const uint32_t both = block_or2((nz != 0 ? 1u : 0u) | (ovf != 0 ? 2u : 0u), or_slots);
  if constexpr (TRACK) {
    if (lt == 0) {
      const int n = nnz_in + dsum;
      a.done_step[g] = n;
      a.done[g] = n == 0 ? 1 : 0;
      if (a.overflow && (both & 2u)) a.overflow[g] = 1;
    }
    return;
  }
  if (lt == 0) {
    a.done[g] = (both & 1u) ? 0 : 1;
    if (a.overflow && (both & 2u)) a.overflow[g] = 1;
  }
}


// =============================================================================================
// tg_step_tracked_i8 at S = 25 (round 3): the in-place step that reads ONLY what the action touches.
// The zero test is what makes tg_step_i8 read every byte of a game; with the number of non-zero entries CARRIED per game
// (nnz: exact on entry, updated here) the step needs the chunks whose rows have u_i v_j != 0 -- ~14 % of them under the
// reference's factor distribution, in ~25 % of the game's 128-byte lines (a row block i with u_i = 0 is never touched) --
// and done = (nnz == 0).  The price is a second, dependent memory round trip (tokens -> which chunks -> those chunks),
// which the resident workgroups of a CU hide from each other.
// Structure of s25_step_kernel: tokens -> the w windows (F) and the row products (uvt) -> barrier -> every thread
// classifies its four chunk INDICES (no data yet) -> candidates into the wavefront's queue -> dense pass: lane k LOADS
// the chunk of entry k, counts its non-zero bytes, applies the action (chunk arithmetic of s25_step_kernel), counts
// again, stores when changed -> the workgroup's nnz delta and overflow flag through LDS.
// =============================================================================================
// One WAVEFRONT per game (four games per workgroup, no workgroup barrier: every table is written and read by its own
// wavefront, and LDS serves one wavefront's accesses in order).  With a workgroup per game the launch was bound by two
// dependent round trips times two rounds of workgroups (14.6 us at 4 096 games, no better than the full step); a
// wavefront per game keeps 8 192 games resident, so BASELINE config 5's share is ONE round.
// The queue holds 192 entries (three per lane: the loads of a dense pass are all in flight together -- with one entry per
// lane and a pass per 64 candidates the ~140 candidates of a game cost three dependent round trips); it is flushed (a
// dense pass) whenever the next batch of candidates would not fit.
__global__ __launch_bounds__(kBlock, 7) void s25_tracked_kernel(ApplyArgs a, int32_t* nnz) {
  constexpr int S = 25, NW = kBlock / 64;
  using G = PGeo<S, kBlock>;
  // queue entries per lane / per wavefront: the loads of a dense pass are all in flight together.  (Measured with 1 / 2 /
  // 3 per lane at 4 096, 32 768 and 139 264 games: a loop with one entry per lane and a flush per 64 candidates 13.0 /
  // 59.6 / 289 us, three per lane 12.9 / 64.5 / 296 us -- and an unrolled classification with in-line flushes for 1 or 2 per
  // lane 20 / 129 / 473 us: the flush must stay the rare case, its code out of the instruction stream.)
  constexpr int QPL = 3, QCAP = 64 * QPL;
  constexpr int WLEN = 2 * G::WE, NCL = (G::NCHUNK + 63) / 64, NRW = (S * S + 63) / 64;
  __shared__ __attribute__((aligned(16))) short Fw[NW][(WLEN + 7) & ~7];  // two periodic copies of w (the second shifted by one)
  __shared__ __attribute__((aligned(16))) int4 qm[NW][QCAP];              // (chunk index, uv of its first row, of its second row, -)
  __shared__ __attribute__((aligned(8))) int uvt[NW][S * S + 3];          // -u_i v_j per row (i, j), 0 behind the last
  __shared__ __attribute__((aligned(4))) int8_t tokb[NW][3 * S + 1];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  int64_t g = static_cast<int64_t>(blockIdx.x) * NW + wave;
  const bool live = g < a.B;
  if (!live) g = a.B - 1;  // a dead wavefront shadows the last game; nothing of it is stored
  const int8_t* const tok = a.actions + g * (3 * S);
  int8_t* const st = a.out + g * a.out_stride;
  const int nnz_in = nnz[g];  // (requested now: its round trip must not come behind the last dense pass)
  // ---- the game's 75 tokens -> LDS; tables from there ----
  tokb[wave][lane] = tok[lane];
  if (lane + 64 < 3 * S) tokb[wave][lane + 64] = tok[lane + 64];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int q = lane + 64 * k;
    if (q < WLEN) {
      const int copy = q >= G::WE;
      const int qq = q - copy * G::WE;
      Fw[wave][q] = static_cast<short>(tokb[wave][2 * S + (qq + copy) % S] - a.shift);
    }
  }
#pragma unroll
  for (int k = 0; k < NRW; ++k) {
    const int row = lane + 64 * k;
    if (row < S * S) {
      const int i = row / S, j = row - i * S;
      uvt[wave][row] = mul24_pinned(a.shift - tokb[wave][i], tokb[wave][S + j] - a.shift);  // -u_i v_j
    }
  }
  if (lane == 0) uvt[wave][S * S] = 0;
  __builtin_amdgcn_wave_barrier();

  int delta = 0;
  uint32_t ovf = 0;
  // one candidate chunk (already loaded): count, apply, count, store
  auto touch = [&](int c, int uv0, int uv1, const uint4& x) {
    const bool tail = G::TAIL != 0 && c == G::NCHUNK - 1;
    const int l0 = (16 * c) % S, hi = S - l0;  // elements k < hi belong to the window's first row
    const uint32_t* wp = reinterpret_cast<const uint32_t*>(Fw[wave] + (l0 & 1) * G::WE + (l0 & ~1));
    uint32_t wraw[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) wraw[p] = wp[p];
    const int c0 = max(-32767, min(32767, uv0)), c1 = max(-32767, min(32767, uv1));
    const uint32_t pr0 = __builtin_amdgcn_perm(static_cast<uint32_t>(c0), static_cast<uint32_t>(c0), 0x05040100u);
    const uint32_t pr1 = __builtin_amdgcn_perm(static_cast<uint32_t>(c1), static_cast<uint32_t>(c1), 0x05040100u);
    uint32_t A[8];
    unpack_pairs(x, A);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const uint32_t pr = (2 * p + 1 < hi) ? pr0 : ((2 * p >= hi) ? pr1 : __builtin_amdgcn_perm(pr1, pr0, 0x07060100u));
      A[p] = pk_mad_i16_sat(pr, wraw[p], A[p]);
    }
    uint32_t c16 = 0, cnz = 0;
    uint4 res = pack_pairs(A, cnz, c16);
    if (__builtin_expect((c16 & 0xFF00FF00u) != 0, 0)) {  // rare: the chunk again in 32 bits (wrapped bytes + flag)
      const uint32_t pd[4] = {x.x, x.y, x.z, x.w};
      uint32_t rd[4];
      int o32 = 0;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        int e[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int k = 4 * d + t;
          const int w = static_cast<short>(wraw[k >> 1] >> (16 * (k & 1)));
          e[t] = sbyte(pd[d], t) + (k < hi ? uv0 : uv1) * w;
          o32 |= e[t] + 128;
        }
        rd[d] = pack4(e[0], e[1], e[2], e[3]);
      }
      res = uint4{rd[0], rd[1], rd[2], rd[3]};
      ovf |= static_cast<uint32_t>(o32) & ~255u;
    }
    if (tail) {  // bytes past the tensor: zero on entry (load_chunk), their rows i >= S add nothing; keep them out of the count
      res.z &= 0x000000FFu;
      res.w = 0;
    }
    delta += nz_bytes16(res) - nz_bytes16(x);
    if (live && (res.x != x.x || res.y != x.y || res.z != x.z || res.w != x.w)) store_chunk<G::TAIL>(st + 16 * c, res, tail);
  };
  auto dense_pass = [&](int n) {  // n entries of the queue: lane k takes entries k, k + 64, ...; ALL their loads first
    __builtin_amdgcn_wave_barrier();
    int4 me[QPL];
    uint4 x[QPL];
#pragma unroll
    for (int k = 0; k < QPL; ++k) {
      const int e = lane + 64 * k;
      me[k] = qm[wave][e < n ? e : 0];
      if (e >= n) me[k].x = -1;
    }
#pragma unroll
    for (int k = 0; k < QPL; ++k) {
      const int c = me[k].x < 0 ? 0 : me[k].x;
      x[k] = load_chunk<G::TAIL>(st + 16 * c, G::TAIL != 0 && c == G::NCHUNK - 1);
    }
#pragma unroll
    for (int k = 0; k < QPL; ++k)
      if (me[k].x >= 0) touch(me[k].x, me[k].y, me[k].z, x[k]);
    __builtin_amdgcn_wave_barrier();
  };
  // ---- which chunks does the action touch?  (indices only.)  All sixteen classifications first -- their table reads in
  // flight together -- then the ballots; the queue is flushed when the next batch would not fit (dense factors only).
  // (only the candidate BITS are kept across the two stages -- the products of a candidate are read again when it is
  // queued; thirty-two of them in registers spilled)
  auto rows_of = [&](int n, int& uv0, int& uv1) {
    const int c = lane + 64 * n;
    const bool cv = c < G::NCHUNK;
    const int cc = cv ? c : 0;
    const int r0 = (16 * cc) / S, l0 = 16 * cc - S * r0;
    uv0 = cv ? uvt[wave][r0] : 0;
    uv1 = (cv && l0 + 16 > S) ? uvt[wave][r0 + 1] : 0;
  };
  uint32_t cbits = 0;
#pragma unroll
  for (int n = 0; n < NCL; ++n) {
    int uv0, uv1;
    rows_of(n, uv0, uv1);
    cbits |= ((uv0 | uv1) != 0 ? 1u : 0u) << n;
  }
  int total = 0;  // wave-uniform
#pragma unroll 1
  for (int n = 0; n < NCL; ++n) {
    const bool cand = (cbits >> n) & 1u;
    const unsigned long long m = __ballot(cand);
    const int cnt = __builtin_popcountll(m);
    if (total + cnt > QCAP) {
      dense_pass(total);
      total = 0;
    }
    const int slot = total + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)));
    if (cand) {
      int uv0, uv1;
      rows_of(n, uv0, uv1);
      qm[wave][slot] = int4{lane + 64 * n, uv0, uv1, 0};
    }
    total += cnt;
  }
  dense_pass(total);
  // ---- the game's nnz and flags ----
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
// S = 9 single step (3x3 matrix multiplication), wavefront-local: four games per wavefront, a team of 16 lanes per
// game, ALL 16 lanes busy (chunks lt, lt + 16, lt + 32 of the game's 46; packed_kernel<9, 16> keeps 9 of 16 lanes busy
// with 6 chunks each for its period trick, stages through LDS with workgroup barriers and scans the tokens first).
// No workgroup barrier at all: a team's int16 table (packed_kernel's: -u, 0, v, pad, two periodic copies of w) is
// written and read by its own wavefront, and LDS serves one wavefront's accesses in order.
// Arithmetic only where the action acts (see s16_step_kernel): a chunk touches up to three rows (i, j); it is a
// candidate when one of them has u_i v_j != 0 (~23 %).  Candidates of the wavefront's four games go through one
// 64-entry queue; in the dense pass a lane reads the window of ITS entry (team table + the chunk's position in the
// row), splits it at the two row boundaries with masks from a small table ("elements below h"), does the
// saturating int16 multiply-adds (32-bit redo when the range test fails: exact or flagged, as everywhere), stores.
// done / overflow per game: ballots -- the owners' over the unchanged chunks, the dense lanes' per team.
// =============================================================================================
//
// MODE == EXPAND (tg_expand_i8): the same kernel with one team per CHILD g = parent * k + c -- "a step of the parent's
// state with the child's action, written to slot g", as s4_expand_kernel -- plus the `changed` byte.  All 16 lanes busy
// and arithmetic only on the candidate chunks, where packed_kernel<9,16,EXPAND> keeps 9 of 16 lanes busy and does the
// whole unpack / multiply-add / pack for every chunk of every child (B = 32 768, k = 8: 80 us; this form 61 us).
template <int MODE>
__global__ __launch_bounds__(kBlock, 6) void s9_step_kernel(ApplyArgs a) {  // (14 spills at 64 VGPRs)
  static_assert(MODE == STEP || MODE == EXPAND, "s9_step_kernel: single step or expand");
  constexpr int S = 9, TS = 16;
  using G = PGeo<S, TS>;  // (UVLEN, WE, FSTRIDE, NCHUNK, TAIL as in packed_kernel<9, 16>)
  static_assert(G::FSTRIDE <= 5 * TS && G::NCHUNK <= 3 * TS && G::TAIL == 9, "s9_step_kernel geometry");
  constexpr int QCAP = 64, GPW = 64 / TS, NWAVE = kBlock / 64;
  __shared__ __attribute__((aligned(16))) short F[NWAVE * GPW][G::FSTRIDE];
  __shared__ __attribute__((aligned(16))) uint4 qd[NWAVE][QCAP];
  __shared__ __attribute__((aligned(16))) int4 qm[NWAVE][QCAP];   // (chunk | team << 8, uv of the three rows)
  __shared__ __attribute__((aligned(16))) uint32_t below[17][8];  // below[h][p]: int16-pair mask of window elements k < h
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = lane >> 4, lt = lane & 15;
  int64_t g = (static_cast<int64_t>(blockIdx.x) * NWAVE + wave) * GPW + team;
  const int64_t ngames = MODE == EXPAND ? a.B * a.nact : a.B;  // (EXPAND: below 2^31, checked on the host)
  const bool live = g < ngames;
  if (!live) g = ngames - 1;
  const int8_t* const tok = a.actions + g * (3 * S);
  const int64_t gin = MODE == EXPAND ? static_cast<int64_t>(static_cast<uint32_t>(g) / static_cast<uint32_t>(a.nact)) : g;
  const int8_t* const in = a.in + gin * a.in_stride;
  int8_t* const out = a.out + g * a.out_stride;
  short* const Fg = F[wave * GPW + team];
  // ---- loads: the lane's table entries (tokens) first, then its three chunks ----
  int tv[5], sidx[5];
#pragma unroll
  for (int e = 0; e < 5; ++e) {
    const int pos = lt + TS * e;
    int si = -1;
    if (pos < S) {
      si = pos;                                   // u (stored negated)
    } else if (pos > S && pos <= 2 * S) {
      si = S + (pos - S - 1);                     // v
    } else if (pos >= G::UVLEN && pos < G::FSTRIDE) {
      int q = pos - G::UVLEN;
      const int copy = q >= G::WE;
      q -= copy * G::WE;
      si = 2 * S + (q + copy) % S;                // periodic copies of w (copy 1 shifted by one)
    }
    sidx[e] = si;
    tv[e] = tok[si >= 0 ? si : 0];
  }
  // whole 16-byte chunks; the last chunk's bytes past S^3 = 729 lie inside the game's stride (736) except, possibly,
  // for the LAST game of the batch, which takes load_chunk's byte-wise tail
  const bool last_game = gin == a.B - 1;
  auto load = [&](int c) {
    uint4 q = uint4{0, 0, 0, 0};
    if (c < G::NCHUNK) q = (c == G::NCHUNK - 1 && last_game) ? load_chunk<G::TAIL>(in + 16 * c, true) : *reinterpret_cast<const uint4*>(in + 16 * c);
    return q;
  };
  uint4 p0 = load(lt), p1 = load(lt + TS), p2 = load(lt + 2 * TS);
  if (lt + 2 * TS == G::NCHUNK - 1) {  // the tail chunk: only 9 bytes belong to the game
    p2.z &= 0x000000FFu;
    p2.w = 0;
  }
  for (int e = tid; e < 17 * 8; e += kBlock) {
    const int h = e >> 3, p = e & 7;
    below[h][p] = ((2 * p < h) ? 0x0000FFFFu : 0u) | ((2 * p + 1 < h) ? 0xFFFF0000u : 0u);
  }
#pragma unroll
  for (int e = 0; e < 5; ++e) {
    const int pos = lt + TS * e;
    if (pos < G::FSTRIDE) {
      int val = sidx[e] >= 0 ? tv[e] - a.shift : 0;
      if (pos < S) val = -val;
      Fg[pos] = static_cast<short>(val);
    }
  }
  __syncthreads();  // (only for `below`, written once per workgroup; the tables are wavefront-local)
  const bool inplace = MODE == STEP && a.in == a.out;
  uint32_t nz = 0, ovf = 0;
  auto differs = [](const uint4& x, const uint4& y) { return x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w; };

  // ---- which chunks does the action touch?  candidates -> the wavefront's queue ----
  int total = 0;  // uniform
  auto enqueue = [&](int n, const uint4& pn, int (&uv)[3]) -> unsigned long long {
    const int c = lt + TS * n;
    const bool cv = c < G::NCHUNK;
    const int r0 = (16 * c) / S, l0 = (16 * c) % S;
#pragma unroll
    for (int sgm = 0; sgm < 3; ++sgm) {
      const int row = r0 + sgm;
      int i = row / S;
      const int j = row - i * S;
      if (!cv || i >= S) i = S;  // Fg[S] == 0: rows past the tensor (and idle chunks) add nothing
      uv[sgm] = mul24_pinned(Fg[i], Fg[S + 1 + j]);
    }
    if (l0 + 16 <= 2 * S) uv[2] = 0;  // the window does not reach a third row
    const bool cand = (uv[0] | uv[1] | uv[2]) != 0;
    const unsigned long long m = __ballot(cand);
    const int slot = total + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)));
    if (cand) {
      if (slot < QCAP) {
        qd[wave][slot] = pn;
        qm[wave][slot] = int4{c | (team << 8), uv[0], uv[1], uv[2]};
      }
    } else {
      nz |= pn.x | pn.y | pn.z | pn.w;
      if (!inplace && cv && live) store_chunk<G::TAIL>(out + 16 * c, pn, c == G::NCHUNK - 1);
    }
    total += __builtin_popcountll(m);
    return m;
  };
  int uva[3], uvb[3], uvc[3];
  const unsigned long long ma = enqueue(0, p0, uva);
  const unsigned long long mb = enqueue(1, p1, uvb);
  const unsigned long long mc = enqueue(2, p2, uvc);

  // one chunk: x + (uv of its row) * w over the window at the chunk's position, table Ft, game base ot
  uint32_t dnz = 0;
  auto finish = [&](const uint4& x, int c, const short* Ft, int8_t* ot, bool lv, int u0, int u1, int u2) {
    const int l0 = (16 * c) % S, h0 = S - l0, h1 = 2 * S - l0;  // elements k < h0: first row; h0 <= k < h1: second
    const uint32_t* wp = reinterpret_cast<const uint32_t*>(Ft + G::UVLEN + (l0 & 1) * G::WE + (l0 & ~1));
    const uint32_t* m0 = below[h0];
    const uint32_t* m1 = below[h1 < 16 ? h1 : 16];
    const int c0 = max(-32767, min(32767, u0)), c1 = max(-32767, min(32767, u1)), c2 = max(-32767, min(32767, u2));
    const uint32_t q0 = __builtin_amdgcn_perm(static_cast<uint32_t>(c0), static_cast<uint32_t>(c0), 0x05040100u);
    const uint32_t q1 = __builtin_amdgcn_perm(static_cast<uint32_t>(c1), static_cast<uint32_t>(c1), 0x05040100u);
    const uint32_t q2 = __builtin_amdgcn_perm(static_cast<uint32_t>(c2), static_cast<uint32_t>(c2), 0x05040100u);
    uint32_t A[8], wraw[8];
    unpack_pairs(x, A);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      wraw[p] = wp[p];
      const uint32_t w0 = wraw[p] & m0[p], t1 = wraw[p] & m1[p];
      A[p] = pk_mad_i16_sat(q0, w0, A[p]);
      A[p] = pk_mad_i16_sat(q1, t1 ^ w0, A[p]);
      A[p] = pk_mad_i16_sat(q2, wraw[p] ^ t1, A[p]);
    }
    uint32_t c16 = 0, cnz = 0;
    uint4 res = pack_pairs(A, cnz, c16);
    if (__builtin_expect((c16 & 0xFF00FF00u) != 0, 0)) {  // rare: the chunk again in 32 bits (wrapped bytes + flag)
      const uint32_t pd[4] = {x.x, x.y, x.z, x.w};
      uint32_t rd[4];
      int o32 = 0;
      cnz = 0;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        int e[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int k = 4 * d + t;
          const int w = static_cast<short>(wraw[k >> 1] >> (16 * (k & 1)));
          e[t] = sbyte(pd[d], t) + (k < h0 ? u0 : (k < h1 ? u1 : u2)) * w;
          o32 |= e[t] + 128;
        }
        rd[d] = pack4(e[0], e[1], e[2], e[3]);
        cnz |= rd[d];
      }
      res = uint4{rd[0], rd[1], rd[2], rd[3]};
      ovf |= static_cast<uint32_t>(o32) & ~255u;
    }
    // (the tail chunk needs no care here: its bytes past the tensor are zero on entry and belong to rows i >= S,
    // whose product is the table's zero entry, so they stay zero; store_chunk writes its 9 bytes)
    dnz |= cnz;
    if (lv && (!inplace || differs(res, x))) store_chunk<G::TAIL>(ot + 16 * c, res, c == G::NCHUNK - 1);
  };
  // One round when the candidates fit the queue; otherwise (dense factors) three rounds, round n taking chunk n of
  // every lane (at most 64 entries by construction) -- the same dense pass either way.  A lane may finish chunks of
  // different games in different rounds: their non-zero / overflow bits are kept per team.
  uint32_t dnz_t = 0, ovf_t = 0;  // bit t: a chunk of team t finished by this lane is non-zero / overflowed
  const int rounds = total <= QCAP ? 1 : 3;  // uniform
  for (int rd = 0; rd < rounds; ++rd) {
    int cnt = total;
    if (rounds == 3) {
      const unsigned long long m = rd == 0 ? ma : (rd == 1 ? mb : mc);
      const uint4 pn = rd == 0 ? p0 : (rd == 1 ? p1 : p2);
      const int u0 = rd == 0 ? uva[0] : (rd == 1 ? uvb[0] : uvc[0]), u1 = rd == 0 ? uva[1] : (rd == 1 ? uvb[1] : uvc[1]),
                u2 = rd == 0 ? uva[2] : (rd == 1 ? uvb[2] : uvc[2]);
      __builtin_amdgcn_wave_barrier();
      if ((u0 | u1 | u2) != 0) {
        const int slot = static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)));
        qd[wave][slot] = pn;
        qm[wave][slot] = int4{(lt + TS * rd) | (team << 8), u0, u1, u2};
      }
      cnt = __builtin_popcountll(m);
    }
    __builtin_amdgcn_wave_barrier();  // (LDS serves one wavefront's accesses in order)
    if (lane < cnt) {
      const uint4 x = qd[wave][lane];
      const int4 me = qm[wave][lane];
      const int dteam = me.x >> 8;
      const int64_t gd = (static_cast<int64_t>(blockIdx.x) * NWAVE + wave) * GPW + dteam;
      dnz = 0;
      const uint32_t ovf_before = ovf;
      ovf = 0;
      finish(x, me.x & 255, F[wave * GPW + dteam], a.out + (gd < ngames ? gd : ngames - 1) * a.out_stride, gd < ngames, me.y, me.z, me.w);
      if (dnz) dnz_t |= 1u << dteam;
      if (ovf) ovf_t |= 1u << dteam;
      ovf = ovf_before;
    }
  }
  // ---- done / overflow per game: the owners' ballot over the unchanged chunks (team = 16-lane slice) and the
  // finishing lanes' per-team bits ----
  const unsigned long long own = __ballot(nz != 0);
  bool any_nz = ((own >> (16 * team)) & 0xFFFFull) != 0, any_ovf = false;
#pragma unroll
  for (int t = 0; t < GPW; ++t) {
    const bool tnz = __ballot((dnz_t >> t) & 1u) != 0, tov = __ballot((ovf_t >> t) & 1u) != 0;
    if (t == team) {
      any_nz = any_nz || tnz;
      any_ovf = tov;
    }
  }
  if constexpr (MODE == EXPAND) {
    if (a.changed) {  // null action <=> u, v or w is the zero vector (the team's table: -u | 0 | v | pad | w ...)
      const bool mine = lt < S;
      const unsigned long long bu = __ballot(mine && Fg[lt] != 0), bv = __ballot(mine && Fg[S + 1 + lt] != 0),
                               bw = __ballot(mine && Fg[G::UVLEN + lt] != 0);
      const int sh = 16 * team;
      const bool nonnull = ((bu >> sh) & 0xFFFFull) != 0 && ((bv >> sh) & 0xFFFFull) != 0 && ((bw >> sh) & 0xFFFFull) != 0;
      if (lt == 0 && live) a.changed[g] = nonnull ? 1 : 0;
    }
  }
  if (lt == 0 && live) {
    a.done[g] = any_nz ? 0 : 1;
    if (a.overflow && any_ovf) a.overflow[g] = 1;
  }
}
