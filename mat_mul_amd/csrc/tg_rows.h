// Row-per-lane accumulation kernels for odd S (included inside namespace tg by tg_kernels.hip).
//
// For step_many / gen_from_factors the state stays on chip for K actions, so the lane <-> data
// mapping is free: here each lane owns WHOLE ROWS (i,j) of S elements.  Along a row u_i*v_j is
// one scalar and the weights w[0..S) are the same for every lane, so one action costs, per row,
// two LDS reads (u_i, v_j), one v_mul_i32_i24, one v_perm_b32 and ceil(S/2) v_pk_mad_i16 -- no
// masks, no per-lane weight windows.  (S = 25: 45 VALU ops per 75 MACs, against 128 per 64 in
// the chunk-per-lane packed kernel.)  The price is a transposition through LDS between the
// 16-byte-chunk layout used for coalesced global access and the row layout, paid once per call
// and amortised over the K actions -- which is why the single step and expand stay on the
// chunk-per-lane kernel.
//
// int16 accumulation is exact under the same condition as tg_packed.h (nact * f^3 <= 32000),
// decided by the same prescan; larger factors run the exact byte-wise form.

//
// SPARSITY.  The rows are dealt so that one wavefront "slot" holds the rows of IPS = 64/S whole
// slices i (S = 25: two slices, 50 lanes).  u_i is then the same for all rows of a slice, and with
// the reference's factor distribution (P(0) = 0.7, datasets.py:31) every u_i of a slot is zero half
// of the time: a wave-uniform __ballot skips that slot's MACs for the action entirely.
template <int S, int TS>
struct RGeo {
  static constexpr int N = S * S * S;
  static constexpr int NROW = S * S;
  static constexpr int NW = TS / 64;                          // wavefronts per team
  static constexpr int IPS = 64 / S;                          // slices per wavefront slot
  static constexpr int NR = (S + IPS * NW - 1) / (IPS * NW);  // slots (= rows) per lane
  static constexpr int NP = (S + 1) / 2;           // int16 pairs per row
  static constexpr int GPB = kBlock / TS;
  static constexpr int NCHUNK = (N + 15) / 16;
  static constexpr int STATE_BYTES = NCHUNK * 16;  // per game, in LDS
  // per-action table (bytes): NP dwords of w pairs, then u[S], 0, v[S] as int16; padded to 16
  static constexpr int TAB_BYTES = ((4 * NP + 2 * (2 * S + 1)) + 15) & ~15;
  static constexpr int ATILE_RAW = 24576 / (GPB * TAB_BYTES);
  static constexpr int ATILE = ATILE_RAW > 64 ? 64 : ATILE_RAW;
  static_assert(ATILE >= 1, "LDS tile");
};

template <int S, int TS, int MODE>
constexpr int rows_lds_bytes(int at) {
  using G = RGeo<S, TS>;
  const int tables = G::GPB * at * G::TAB_BYTES;
  const int state = G::GPB * G::STATE_BYTES;
  const int raw = G::GPB * ((at * 3 * S + 8 + 15) & ~15);
  const int nflag = (MODE == MANY ? TG_MAX_ACTIONS : 0) + 16;  // + one "recompute" byte per team
  return ((tables + state + raw + nflag + 15) & ~15) + 16;     // + the slot array of block_or2
}

template <int S, int TS, int MODE>
__global__ __launch_bounds__(kBlock) void rows_kernel(ApplyArgs a, int flim, int at) {
  static_assert(MODE == MANY || MODE == GENF, "rows_kernel: accumulation modes only");
  using G = RGeo<S, TS>;
  constexpr bool SUB = (MODE == MANY);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int raw_stride = (at * 3 * S + 8 + 15) & ~15;
  uint8_t* const tab_all = smem;
  uint8_t* const state_all = tab_all + G::GPB * at * G::TAB_BYTES;
  int8_t* const raw_all = reinterpret_cast<int8_t*>(state_all + G::GPB * G::STATE_BYTES);
  uint8_t* const flags = reinterpret_cast<uint8_t*>(raw_all + G::GPB * raw_stride);

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
  uint8_t* const tab = tab_all + team * (at * G::TAB_BYTES);
  uint8_t* const st = state_all + team * G::STATE_BYTES;

  auto load_raw = [&](int a0, int na, int& head) {  // tokens of actions [a0,a0+na) -> LDS, unchecked
    (void)load_tokens_checked<TS>(tok + a0 * (3 * S), na * 3 * S, raw, lt, a.shift, 0x7fffffff, head);
  };

  int head0 = 0;
  if (factors_too_large<TS>(tok, a.nact, at, 3 * S, raw, lt, a.shift, flim, head0,
                            reinterpret_cast<uint32_t*>(smem + rows_lds_bytes<S, TS, MODE>(at) - 16))) {
    note_fallback();
    for (int t = 0; t < G::GPB; ++t) {
      const int64_t b = static_cast<int64_t>(blockIdx.x) * G::GPB + t;
      if (b < a.B && flagged_or_all<MODE>(a, b)) slow_game<MODE>(a, b, flags);
    }
    return;
  }

  // ---- lane geometry: slot n of wavefront wv holds slices i = (n*NW + wv)*IPS + [0, IPS) ---------
  const int wv = lt >> 6, ln = lt & 63;
  const int li = ln / S, lj = ln - li * S;  // lane -> (slice within the slot, row j)
  int uoff[G::NR], voff[G::NR], rowb[G::NR];  // byte offsets of u_i, v_j in an action's table; row start in `st`
  bool rv[G::NR];
#pragma unroll
  for (int n = 0; n < G::NR; ++n) {
    const int i = (n * G::NW + wv) * G::IPS + li;
    rv[n] = li < G::IPS && i < S;
    uoff[n] = 4 * G::NP + 2 * (rv[n] ? i : S);  // u[S] == 0: idle lanes accumulate nothing
    voff[n] = 4 * G::NP + 2 * (S + 1 + (rv[n] ? lj : 0));
    rowb[n] = (i * S + lj) * S;
  }

  // ---- state: coalesced 16-byte chunks -> LDS -> rows as int16 pairs -----------------------------
  uint32_t acc[G::NR][G::NP];
#pragma unroll
  for (int n = 0; n < G::NR; ++n)
#pragma unroll
    for (int p = 0; p < G::NP; ++p) acc[n][p] = (MODE == MANY) ? kLatticeZero : 0u;  // MANY: lattice, tg_packed.h
  if constexpr (MODE == MANY) {
    const int8_t* src = a.in + g * a.in_stride;
    for (int c = lt; c < G::NCHUNK; c += TS) {
      uint4 q;
      if (16 * c + 16 <= G::N) {
        q = *reinterpret_cast<const uint4*>(src + 16 * c);
      } else {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int t = 0; 16 * c + t < G::N; ++t)
          w[t >> 2] |= static_cast<uint32_t>(static_cast<uint8_t>(src[16 * c + t])) << (8 * (t & 3));
        q = uint4{w[0], w[1], w[2], w[3]};
      }
      *reinterpret_cast<uint4*>(st + 16 * c) = q;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < G::NR; ++n) {
      if (rv[n]) {
        const int8_t* row = reinterpret_cast<const int8_t*>(st) + rowb[n];
#pragma unroll
        for (int p = 0; p < G::NP; ++p) {
          const int lo = row[2 * p] * 256 + 128;
          const int hi = (2 * p + 1 < S) ? row[2 * p + 1] * 256 + 128 : 128;
          acc[n][p] = __builtin_amdgcn_perm(static_cast<uint32_t>(hi), static_cast<uint32_t>(lo), 0x05040100u);
        }
      }
    }
  }

  // ---- staging: raw tokens -> per-action table { w pairs (dwords) | u[S],0 | v[S] } --------------
  constexpr int ENT = G::NP + 2 * S + 1;  // logical entries per action: NP w pairs, u[S], 0, v[S]
  constexpr int LPA = ENT <= 16 ? 16 : (ENT <= 32 ? 32 : (ENT <= 64 ? 64 : (ENT <= 128 ? 128 : 256)));
  static_assert(LPA <= TS, "one action's table must fit the team");
  constexpr int APP = TS / LPA;  // actions staged per pass
  // what this lane writes is the same for every action: resolve it once
  const int spos = lt % LPA, ksub = lt / LPA;
  int i0 = -1, i1 = -1, dst_off = -1;  // source token indices (-1: zero), destination byte offset
  bool dword = false, neg = false;
  if (spos < G::NP) {
    i0 = 2 * S + 2 * spos;
    i1 = (2 * spos + 1 < S) ? i0 + 1 : -1;
    dst_off = 4 * spos;
    dword = true;
  } else if (spos < G::NP + S) {
    i0 = spos - G::NP;
    neg = SUB;
    dst_off = 4 * G::NP + 2 * (spos - G::NP);
  } else if (spos == G::NP + S) {
    dst_off = 4 * G::NP + 2 * S;
  } else if (spos < ENT) {
    const int j = spos - G::NP - S - 1;
    i0 = S + j;
    dst_off = 4 * G::NP + 2 * (S + 1 + j);
  }
  auto stage = [&](int a0, int na, bool loaded) {
    int head = head0;
    if (!loaded) {
      __syncthreads();
      load_raw(a0, na, head);
    }
    __syncthreads();
    if (dst_off >= 0) {
      for (int k = ksub; k < na; k += APP) {
        const int8_t* t = raw + head + k * (3 * S);
        uint8_t* T = tab + k * G::TAB_BYTES + dst_off;
        int v0 = i0 >= 0 ? t[i0] - a.shift : 0;
        int v1 = i1 >= 0 ? t[i1] - a.shift : 0;
        if (neg) v0 = -v0;
        if (MODE == MANY && dword) {  // lattice weights
          v0 *= 256;
          v1 *= 256;
        }
        if (dword)
          *reinterpret_cast<uint32_t*>(T) =
              __builtin_amdgcn_perm(static_cast<uint32_t>(v1), static_cast<uint32_t>(v0), 0x05040100u);
        else
          *reinterpret_cast<short*>(T) = static_cast<short>(v0);
      }
    }
    __syncthreads();
  };

  uint32_t ovf = 0;
  int done_step = -1;
  uint32_t nzs[G::NR];
#pragma unroll
  for (int n = 0; n < G::NR; ++n) {
    nzs[n] = 0;
#pragma unroll
    for (int p = 0; p < G::NP; ++p) nzs[n] |= acc[n][p];
  }
  if constexpr (MODE == MANY && TS == 256) {
    for (int k = tid; k < a.nact; k += kBlock) flags[k] = 0;
  }
  for (int a0 = 0; a0 < a.nact; a0 += at) {
    const int na = min(at, a.nact - a0);
    stage(a0, na, a.nact <= at);
    for (int k = 0; k < na; ++k) {
      const uint8_t* T = tab + k * G::TAB_BYTES;
      uint32_t wp[G::NP];
#pragma unroll
      for (int p = 0; p < G::NP; ++p) wp[p] = reinterpret_cast<const uint32_t*>(T)[p];
      // every LDS read of the action is issued up front (one latency); only arithmetic is conditional
      int ui[G::NR], vj[G::NR];
#pragma unroll
      for (int n = 0; n < G::NR; ++n) {
        ui[n] = *reinterpret_cast<const short*>(T + uoff[n]);
        vj[n] = *reinterpret_cast<const short*>(T + voff[n]);
      }
#pragma unroll
      for (int n = 0; n < G::NR; ++n) {
        if (__ballot(ui[n] != 0) == 0) continue;  // every u_i of this slot is zero: wave-uniform skip
        const int uv = mul24_pinned(ui[n], vj[n]);
        const uint32_t pr = __builtin_amdgcn_perm(static_cast<uint32_t>(uv), static_cast<uint32_t>(uv), 0x05040100u);
        uint32_t o = 0;
#pragma unroll
        for (int p = 0; p < G::NP; ++p) {
          acc[n][p] = (MODE == MANY) ? pk_mad_i16_sat(pr, wp[p], acc[n][p]) : pk_mad_i16(pr, wp[p], acc[n][p]);
          o |= acc[n][p];
        }
        nzs[n] = o;  // OR of the slot's lattice values, refreshed only when the slot changed
      }
      if constexpr (MODE == MANY) {
        uint32_t nz = 0;
#pragma unroll
        for (int n = 0; n < G::NR; ++n) nz |= nzs[n];
        if constexpr (TS == 256) {
          if (nz & 0xFF00FF00u) flags[a0 + k] = 1;
        } else {
          if (!team_any<TS>((nz & 0xFF00FF00u) != 0) && done_step < 0) done_step = a0 + k;
        }
      }
    }
  }

  // ---- MANY: off the lattice <=> some step overflowed int8 -> exact recompute, nothing stored -------
  if constexpr (MODE == MANY) {
    uint32_t off = 0;
#pragma unroll
    for (int n = 0; n < G::NR; ++n)
#pragma unroll
      for (int p = 0; p < G::NP; ++p) off |= (acc[n][p] ^ kLatticeZero) & 0x00FF00FFu;
    if constexpr (TS == 256) {
      if (__syncthreads_or(off != 0)) {
        note_fallback();
        if (live) slow_game<MODE>(a, g, flags);
        return;
      }
    } else {
      const bool bad = team_any<TS>(off != 0);
      uint8_t* const badF = flags + TG_MAX_ACTIONS;
      __syncthreads();
      if (lt == 0) badF[team] = bad && live;
      __syncthreads();
      for (int t = 0; t < G::GPB; ++t)
        if (badF[t]) {
          note_fallback();
          slow_game<MODE>(a, static_cast<int64_t>(blockIdx.x) * G::GPB + t, flags);
        }
      if (bad) return;
    }
  }

  // ---- rows -> LDS bytes -> coalesced 16-byte stores ----------------------------------------------
  __syncthreads();
#pragma unroll
  for (int n = 0; n < G::NR; ++n) {
    if (rv[n]) {
      uint8_t* row = st + rowb[n];
#pragma unroll
      for (int p = 0; p < G::NP; ++p) {
        if constexpr (MODE == MANY) {
          row[2 * p] = static_cast<uint8_t>(acc[n][p] >> 8);
          if (2 * p + 1 < S) row[2 * p + 1] = static_cast<uint8_t>(acc[n][p] >> 24);
        } else {
          ovf |= pk_add_u16(acc[n][p], 0x00800080u);
          row[2 * p] = static_cast<uint8_t>(acc[n][p]);
          if (2 * p + 1 < S) row[2 * p + 1] = static_cast<uint8_t>(acc[n][p] >> 16);
        }
      }
    }
  }
  __syncthreads();
  if (live) {
    int8_t* dst = a.out + g * a.out_stride;
    for (int c = lt; c < G::NCHUNK; c += TS) {
      if (16 * c + 16 <= G::N) {
        *reinterpret_cast<uint4*>(dst + 16 * c) = *reinterpret_cast<const uint4*>(st + 16 * c);
      } else {
        for (int t = 0; 16 * c + t < G::N; ++t) dst[16 * c + t] = static_cast<int8_t>(st[16 * c + t]);
      }
    }
  }
  if constexpr (MODE == MANY) {
    if constexpr (TS == 256) {
      if (tid == 0) {
        for (int k = 0; k < a.nact; ++k)
          if (!flags[k]) { done_step = k; break; }
      }
    }
    if (lt == 0 && live) a.done_step[g] = done_step;
  } else {
    bool any_ovf;
    if constexpr (TS == 256) {
      any_ovf = __syncthreads_or((ovf & 0xFF00FF00u) != 0);
    } else {
      any_ovf = team_any<TS>((ovf & 0xFF00FF00u) != 0);
    }
    if (lt == 0 && live && a.overflow && any_ovf) a.overflow[g] = 1;
  }
}
