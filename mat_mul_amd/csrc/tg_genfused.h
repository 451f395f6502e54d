// tg_genfused.h -- the synthetic-demonstration generator in ONE kernel (included by tg_kernels.hip after tg_mfma.h).
//
// create_synthetic_demo (utils.py:203-233) / _create_synthetic_demos (datasets.py:124-142) for S = 9, 16, 25:
//   Philox -> factor bytes in registers -> [change of basis: one int8 MFMA per (mode, 32 actions)] -> the transposed
//   factor image T[x][r] in LDS, the tokens straight to global memory (round 3) -> the accumulation tiles of
//   tg_mfma.h -> the target image -> target out.
// Round 1 ran three kernels (gen_tokens_kernel, basis_tokens_mfma_kernel, genf_mfma_kernel): the tokens made a round
// trip through memory twice (19.7 MB written, transformed in place, re-read at S=25, R=64, B=4096) and the token
// kernel burnt a 32-bit Philox lane per 3-way draw.  Here the factors never leave the chip before they are final:
// per demo the kernel writes S^3 + 3SR bytes and reads nothing (with a basis: 3 S^2 bytes).
//
// Draw phase, one job = (mode x, 32 actions) per wavefront pass.  The lane mapping IS the int8 MFMA B-fragment
// mapping -- lane (col, h) owns elements k = 16h .. 16h+15 of the vector of action r0 + col -- so the change of basis
// D[a][r] = sum_i M_x[a][i] f_r[i] takes the drawn bytes as they stand (A fragment = row `col` of M_x, read from
// global memory), and without a basis the same registers go straight to T and -- as tokens -- to global memory.  One Philox block
// is eight 16-bit draws evaluated two at a time with packed int16 ops (draw_block16); a lane runs the blocks of its
// half (S = 25: blocks 2h, 2h+1; S <= 16: block h, handed to the lower half by v_permlane32_swap).  A vector that
// comes out all zero is redrawn (attempt + 1) by its two lanes; the wavefront loops while any vector needs it
// (P = 0.7^S per vector: 1.3e-4 at S = 25).
#pragma once

struct GenArgs {
  int8_t* target;
  int8_t* actions;
  uint8_t* overflow;
  const int8_t* basis;  // (B,3,S,S) int8, or nullptr
  int64_t B;
  int64_t out_stride;
  uint64_t seed;
  uint64_t gid0;
  int R;
  int shift;
  Dist D;
#ifdef TG_AB_SWITCHES
  int ablate;  // A/B build only (TG_GF_ABLATE): 1 no Philox, 2 no LDS writes of the draw, 4 no tiles, 8 no target store, 16 no token store
#endif
};

#ifdef TG_AB_SWITCHES
#define TG_GF_ON(bit) (!(ga.ablate & (bit)))
#else
#define TG_GF_ON(bit) true
#endif
// T, the target image, the four OR words (no token image: a lane's tokens leave from its registers)
template <int S>
constexpr int genfused_lds_bytes(int Rp) {
  return MGeo<S>::TROWS * (Rp + 16) + MGeo<S>::IMG + 32 + 16;
}

// bytes 16h .. 16h+15 of an S-byte row in global memory (any alignment; nothing past the row is read; bytes >= S are 0).
// As few load instructions as possible (round 3: scattered accesses are paid per instruction): 8-byte pieces while they fit
// the row, single bytes for the rest -- three instructions at S = 25 (8 for both lane halves, 8 for the lower, 1 for the upper)
// where dword pieces took seven.
template <int S>
__device__ __forceinline__ v4i row_fragment16(const int8_t* row, int h) {
  struct __attribute__((packed)) P8 { uint32_t lo, hi; };
  uint32_t w[4] = {0, 0, 0, 0};
  auto half = [&](auto hc) {  // the lane half as a compile-time constant: every piece's validity is known
    constexpr int H = decltype(hc)::value;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      constexpr int k0base = 16 * H;
      const int k0 = k0base + 8 * p;
      if (k0 + 7 < S) {
        const P8 v = *reinterpret_cast<const P8*>(row + k0);
        w[2 * p] = v.lo;
        w[2 * p + 1] = v.hi;
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (k0 + t < S) w[2 * p + (t >> 2)] |= static_cast<uint32_t>(static_cast<uint8_t>(row[k0 + t])) << (8 * (t & 3));
      }
    }
  };
  if (h == 0) half(std::integral_constant<int, 0>{});
  else half(std::integral_constant<int, 1>{});
  return v4i{static_cast<int>(w[0]), static_cast<int>(w[1]), static_cast<int>(w[2]), static_cast<int>(w[3])};
}

// The exact form of one game from its EMITTED tokens in LDS (any factor magnitude): the fallback of gen_fused_kernel
// for games whose transformed factors leave the byte-product range of the matrix-core path.  Whole workgroup.
template <int S, int NTHREADS>
__device__ __forceinline__ int exact_target_from_tokens(const uint8_t* tok, int R, int shift, int8_t* out) {
  constexpr int S2 = S * S, N = S2 * S, A3 = 3 * S;
  int ovf = 0;
  for (int e = threadIdx.x; e < N; e += NTHREADS) {
    const int i = e / S2, rr = e - i * S2, j = rr / S, l = rr - j * S;
    int acc = 0;
    for (int r = 0; r < R; ++r) {
      const int8_t* t = reinterpret_cast<const int8_t*>(tok) + r * A3;
      acc += (t[i] - shift) * (t[S + j] - shift) * (t[2 * S + l] - shift);
    }
    ovf |= (acc < -128) | (acc > 127);
    out[e] = static_cast<int8_t>(acc);
  }
  return ovf;
}

// KS: k-steps of 32 actions known at compile time (1 or 2: R <= 64), 0 = run-time Rp / 32.
//
// Pipeline of one workgroup over its games g0, g1, ... (two barriers per game):
//     draw(g0) | B1 | tiles(g0) | B2 | stores(g0) + draw(g1) | B1 | tiles(g1) | B2 | stores(g1) + draw(g2) | ...
// The stores of a game (LDS images -> global memory, fire and forget) are issued by the wavefronts that have the
// fewest draw jobs of the next game (3 Rp/32 jobs over 4 wavefronts: at R = 64 two wavefronts draw twice, the other two
// draw once and store), so the memory phase hides behind the next game's Philox arithmetic.  The target image and T are
// free again at B1 / B2 respectively.  (Round 2 also kept a double-buffered token image in LDS; since round 3 a lane's
// tokens leave from its registers inside the draw phase by unaligned global stores: the image cost 16 ds_write_b8 + 6 shifts
// per job and lane and a second pass LDS -> global -- 4.1 + 1.0 us of 32.3 by ablation; it was removed in round 4.)
//
// NW wavefronts per workgroup: 4, or 6 when the 3 Rp/32 draw jobs divide by 6 (R = 64: one job per wavefront instead of
// two wavefronts drawing twice while two wait, and 20 tiles as 4+4+3+3+3+3 instead of 5 each).
// CHECK = false (host-proved, BASIS = false only): R * max|value|^3 <= 127, so no target entry can leave int8 and the
// tiles skip their range tracking (the reference's {-1,0,1} with R <= 127).
// TERN: the distribution is the three-valued one (two thresholds), known on the host.
// LUT (round 3; host-proved: TERN, no basis, no CHECK, values exactly (-1, 0, 1)): the u and v rows of T hold the
// ternary codes of lutmul16 (tg_mfma.h) and the tiles look their byte products up -- 8 VALU instructions per 32
// actions and tile instead of 16 (the byte products were ~40 % of this kernel's instructions).
template <int S, int KS, bool BASIS, int NW, bool CHECK = true, bool TERN = false, bool LUT = false>
__global__ __launch_bounds__(64 * NW) void gen_fused_kernel(GenArgs ga, int Rp) {
  static_assert(!LUT || (TERN && !BASIS && !CHECK), "the lookup form is for the plain ternary generator");
  using G = MGeo<S>;
  constexpr int NTHREADS = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) uint8_t mfma_smem[];
  if constexpr (KS != 0) Rp = 32 * KS;
  const int RS = Rp + 16;
  const int R = ga.R, blk = R * G::A3;
  int8_t* const T = reinterpret_cast<int8_t*>(mfma_smem);
  uint8_t* const img = mfma_smem + G::TROWS * RS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  if (static_cast<int64_t>(blockIdx.x) >= ga.B) return;
  if (!TG_GF_ON(32)) return;  // (A/B build: launch + dispatch only)

  // T starts all zero and only cells (x, i < S, r < R) are ever written: W rows l >= S and the padding actions
  // r >= R (up to Rp) stay zero for every game
  for (int e = 16 * tid; e < G::TROWS * RS; e += 16 * NTHREADS) *reinterpret_cast<uint4*>(T + e) = uint4{0, 0, 0, 0};

  TileMap<S, NW> tm;
  make_tile_map(tm, RS, wave, col, h);

  // draw role.  NB Philox blocks of 8 draws per vector.  NB <= 2: lane half h runs block h and the lower half
  // assembles the vector (elements 0..15); NB >= 3: lane half h runs blocks 2h, 2h+1 = its own 16 elements.
  constexpr int NB = (S + 7) / 8;
  constexpr bool kSwap = NB <= 2;
  const int q0 = kSwap ? h : 2 * h;                     // first block of this lane
  const bool draws = kSwap ? (h < NB) : (2 * h < NB);   // does this lane half run Philox at all
  uint32_t vmask[4];                                    // valid bytes of the lane's (up to) 16 drawn elements
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int nv = S - (8 * q0 + 4 * j);  // drawn dword j covers elements 8 q0 + 4 j .. + 3
    vmask[j] = nv >= 4 ? 0xFFFFFFFFu : (nv <= 0 ? 0u : ((1u << (8 * nv)) - 1u));
  }
  const int kbase = kSwap ? 0 : 16 * h;                 // first element of the lane's fragment after assembly
  const bool holds = kSwap ? (h == 0) : true;           // does this lane hold a fragment after assembly
  const uint32_t k0 = static_cast<uint32_t>(ga.seed), k1 = static_cast<uint32_t>(ga.seed >> 32);
  const uint32_t shp = (static_cast<uint32_t>(ga.shift) & 0xFFFFu) | (static_cast<uint32_t>(ga.shift) << 16);
  uint32_t one16 = 0x00010001u, base16 = ga.D.base16;  // VGPR-resident operands of the packed draw evaluation
  asm volatile("" : "+v"(one16), "+v"(base16));
  const int NTR = Rp >> 5, njob = 3 * NTR;
  // store role: the wavefronts with the fewest draw jobs (wave >= njob % NW when the jobs do not divide evenly)
  const int nheavy = njob % NW;                         // wavefronts 0 .. nheavy-1 draw once more than the others
  const bool light = wave >= nheavy;
  const int light_tid = tid - 64 * nheavy, light_threads = NTHREADS - 64 * nheavy;

  // S = 17 .. 31 (round 4): a vector's S token bytes leave as ONE 16-byte store per lane -- the lower lane half (elements
  // 0..15 in Kv) its bytes 0..15, the upper half (elements 16..31) the LAST sixteen bytes S-16 .. S-1: its own S - 16 behind
  // the 32 - S bytes in front of them, which it gets from its partner lane (v_permlane32_swap) and shifts into place
  // (v_alignbyte_b32); the overlap is written twice with the same values.  Scattered stores are paid per INSTRUCTION (a
  // lane's bytes lie 3 S bytes from its neighbour's: ~3 us of a 28 us launch per store instruction per job, round 3);
  // this was 16 | 8 + 1 bytes = three instructions per job at S = 25.  Called by every lane (the exchange is cross-lane).
  auto store_vector16 = [&](const uint32_t (&Kv)[4], int8_t* vec, bool on) {
    constexpr int OFF = S > 16 ? S - 16 : 1, SH = OFF & 3, W0 = OFF >> 2;
    uint32_t Lw[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int w = W0 + t;  // dword w of the vector's 32 bytes: lower half's K[w] for w < 4, the upper half's own K[w - 4]
      if (w < 4) Lw[t] = __builtin_amdgcn_permlane32_swap(0u, Kv[w], false, false)[0];
      else Lw[t] = Kv[(w - 4) & 3];
    }
    uint32_t Dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t up = SH ? __builtin_amdgcn_alignbyte(Lw[d + 1], Lw[d], static_cast<uint32_t>(SH)) : Lw[d];
      Dq[d] = h ? up : Kv[d];
    }
    int8_t* const pq = vec + (h ? OFF : 0);
    const v4u_t q{Dq[0], Dq[1], Dq[2], Dq[3]};
    if (on) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(pq), "v"(q) : "memory");
  };

  // ---- draw (and transform) the factors of game g: registers -> T, the tokens -> global memory ----
  auto draw = [&](int64_t g, int& big, int& bad) {
    if (!TG_GF_ON(64)) return;  // (A/B build: no draw phase at all)
    const uint64_t gid = ga.gid0 + static_cast<uint64_t>(g);
    for (int job = wave; job < njob; job += NW) {
      const int x = job / NTR, r = 32 * (job - x * NTR) + col;
      const bool active = r < R;
      v4i fa;
      if constexpr (BASIS)  // row a = col of M_x (rows >= S shadow the last one; their results are never used)
        fa = row_fragment16<S>(ga.basis + ((g * 3 + x) * S + (col < S ? col : S - 1)) * S, h);
      uint32_t Dw[4] = {0, 0, 0, 0};  // factor bytes of this lane's blocks (two dwords per block)
      uint32_t Kw[4] = {0, 0, 0, 0};  // the same as tokens (value + shift)
      uint32_t Cw[4] = {0, 0, 0, 0};  // LUT: what goes to T (codes for u and v, the factor bytes for w)
      bool need = active && draws;
      uint32_t attempt = 0;
      while (true) {
        if (need && !TG_GF_ON(1)) {
          Dw[0] = 0x01FF0001u & vmask[0];
          Kw[0] = 0x02000102u;
        }
        if (need && TG_GF_ON(1)) {
#pragma unroll
          for (int b = 0; b < (kSwap ? 1 : 2); ++b) {
            if (8 * (q0 + b) < S) {
              const U4 o = philox4x32_10(U4{static_cast<uint32_t>(gid), static_cast<uint32_t>(gid >> 32),
                                            static_cast<uint32_t>(3 * r + x), (attempt << 8) | static_cast<uint32_t>(q0 + b)},
                                         k0, k1);
              const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
              uint32_t P[4];
              draw_block16<TERN>(ow, ga.D, one16, base16, P);
              Dw[2 * b] = __builtin_amdgcn_perm(P[1], P[0], 0x06040200u) & vmask[2 * b];
              Dw[2 * b + 1] = __builtin_amdgcn_perm(P[3], P[2], 0x06040200u) & vmask[2 * b + 1];
              if constexpr (!BASIS) {
                Kw[2 * b] = __builtin_amdgcn_perm(pk_add_u16(P[1], shp), pk_add_u16(P[0], shp), 0x06040200u);
                Kw[2 * b + 1] = __builtin_amdgcn_perm(pk_add_u16(P[3], shp), pk_add_u16(P[2], shp), 0x06040200u);
              }
              if constexpr (LUT) {  // what T gets for this mode: u -> value + 1, v -> 4 m(value), w -> the value
                if (x < 2) {        // (wave-uniform)
                  uint32_t c0 = Kw[2 * b], c1 = Kw[2 * b + 1];  // the tokens ARE value + 1 under the usual shift
                  if (ga.shift != 1) {
                    c0 = __builtin_amdgcn_perm(pk_add_u16(P[1], one16), pk_add_u16(P[0], one16), 0x06040200u);
                    c1 = __builtin_amdgcn_perm(pk_add_u16(P[3], one16), pk_add_u16(P[2], one16), 0x06040200u);
                  }
                  Cw[2 * b] = x == 0 ? c0 : __builtin_amdgcn_perm(0u, kLutCodeV, c0);
                  Cw[2 * b + 1] = x == 0 ? c1 : __builtin_amdgcn_perm(0u, kLutCodeV, c1);
                } else {
                  Cw[2 * b] = Dw[2 * b];
                  Cw[2 * b + 1] = Dw[2 * b + 1];
                }
              }
            }
          }
        }
        // the vector (col) is accepted when either half holds a non-zero element
        const unsigned long long nzm = __ballot((Dw[0] | Dw[1] | Dw[2] | Dw[3]) != 0);
        const uint32_t ok32 = static_cast<uint32_t>(nzm) | static_cast<uint32_t>(nzm >> 32);
        need = need && !((ok32 >> col) & 1u) && (attempt + 1 < (1u << 16));
        if (__ballot(need) == 0) break;
        ++attempt;
      }
      v4i F, K;
      if constexpr (LUT) {  // from here on "the factor bytes" are what T gets
#pragma unroll
        for (int j = 0; j < 4; ++j) Dw[j] = Cw[j];
      }
      if constexpr (kSwap) {  // the upper half's block becomes elements 8..15 of the lower half's fragment
        const auto s0 = __builtin_amdgcn_permlane32_swap(Dw[0], 0u, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(Dw[1], 0u, false, false);
        F = h == 0 ? v4i{static_cast<int>(Dw[0]), static_cast<int>(Dw[1]), static_cast<int>(s0[1]), static_cast<int>(s1[1])}
                   : v4i{0, 0, 0, 0};
        if constexpr (!BASIS) {
          const auto t0 = __builtin_amdgcn_permlane32_swap(Kw[0], 0u, false, false);
          const auto t1 = __builtin_amdgcn_permlane32_swap(Kw[1], 0u, false, false);
          K = v4i{static_cast<int>(Kw[0]), static_cast<int>(Kw[1]), static_cast<int>(t0[1]), static_cast<int>(t1[1])};
        }
      } else {
        F = v4i{static_cast<int>(Dw[0]), static_cast<int>(Dw[1]), static_cast<int>(Dw[2]), static_cast<int>(Dw[3])};
        K = v4i{static_cast<int>(Kw[0]), static_cast<int>(Kw[1]), static_cast<int>(Kw[2]), static_cast<int>(Kw[3])};
      }
      if constexpr (!BASIS) {
        // elements k < KBOTH are valid in both lane halves (kbase 0 and 16), KBOTH <= k < KLOW only for kbase == 0:
        // two straight runs of byte writes instead of sixteen individually predicated ones
        constexpr int KBOTH = kSwap ? 0 : (S - 16 < 16 ? S - 16 : 16), KLOW = S < 16 ? S : 16;
        if (active && holds && TG_GF_ON(2)) {
          // (round 4, tried and taken out again: the four lanes of a quad transposing their 4 x 4 bytes -- two DPP exchanges, two
          // v_perm_b32 per dword -- so that T is written as four aligned dwords instead of sixteen bytes: bit-exact, LDS
          // instructions -15 %, SQ_WAIT_INST_LDS -40 %, but +66 VALU per game and wavefront, SQ_LDS_BANK_CONFLICT unchanged
          // (the conflicts are not these writes) and 25.29 against 25.18 us.)
          int8_t* const tcol = T + (x * S + kbase) * RS + r;
#pragma unroll
          for (int k = 0; k < KBOTH; ++k) tcol[k * RS] = static_cast<int8_t>(static_cast<uint32_t>(F[k >> 2]) >> (8 * (k & 3)));
          if (kbase == 0) {
#pragma unroll
            for (int k = KBOTH; k < KLOW; ++k) tcol[k * RS] = static_cast<int8_t>(static_cast<uint32_t>(F[k >> 2]) >> (8 * (k & 3)));
          }
        }
        if constexpr (!kSwap && S > 16) {
          const uint32_t Kv[4] = {static_cast<uint32_t>(K[0]), static_cast<uint32_t>(K[1]), static_cast<uint32_t>(K[2]), static_cast<uint32_t>(K[3])};
          store_vector16(Kv, ga.actions + g * blk + (3 * r + x) * S, active && TG_GF_ON(16));
        } else if (active && holds && TG_GF_ON(16)) {
          // the vector's tokens: KLOW bytes (lane half 0) / KBOTH bytes (lane half 1) at (3 r + x) S + kbase of the game's
          // block, any alignment: whole dwords as unaligned global stores, then the tail bytes
          // As FEW store instructions and lane transactions as possible: these stores are scattered (a lane's bytes lie 3 S
          // bytes from its neighbour's) and the CU's write path takes them one lane at a time -- one more dword store per job
          // cost 3.7 us of the 28.6 us launch (measured).  So: the widest unaligned pieces (16, 8, 4 bytes), then single bytes.
          int8_t* const gp = ga.actions + g * blk + (3 * r + x) * S + kbase;
          struct __attribute__((packed)) P16 { uint32_t v[4]; };
          struct __attribute__((packed)) P8 { uint32_t v[2]; };
          auto put = [&](int nb) {  // nb: compile-time after inlining (KLOW / KBOTH)
            int o = 0;
            if (nb - o >= 16) {  // (written out: for a byte-aligned address hipcc splits the 16 bytes into two 8-byte stores)
              const v4u_t q{static_cast<uint32_t>(K[0]), static_cast<uint32_t>(K[1]), static_cast<uint32_t>(K[2]), static_cast<uint32_t>(K[3])};
              asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(gp), "v"(q) : "memory");
              o = 16;
            }
            if (nb - o >= 8) { *reinterpret_cast<P8*>(gp + o) = P8{{static_cast<uint32_t>(K[o >> 2]), static_cast<uint32_t>(K[(o >> 2) + 1])}}; o += 8; }
            if (nb - o >= 4) { reinterpret_cast<UnalignedU32*>(gp + o)->v = static_cast<uint32_t>(K[o >> 2]); o += 4; }
            for (; o < nb; ++o) gp[o] = static_cast<int8_t>(static_cast<uint32_t>(K[o >> 2]) >> (8 * (o & 3)));
          };
          if (kbase == 0) put(KLOW);
          else put(KBOTH);
        }
      } else {
        // D[a][r] = sum_i M_x[a][i] f_r[i]: one int8 MFMA; this lane gets a = (t & 3) + 8 (t >> 2) + 4 h of action r
        v16i acc;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0;
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, F, acc, 0, 0, 0);
        if (active && TG_GF_ON(2)) {
          int8_t* const tcol = T + (x * S + 4 * h) * RS + r;
          const int lim = x < 2 ? G::UVLIM : 127, lim_lo = x < 2 ? -G::UVLIM : -128;
          int fmx = 0, fmn = 0;  // range of the emitted factors: ONE test per job instead of five operations per value
          auto emit = [&](int t) {  // register t = row a0 + 4 h, a0 = (t & 3) + 8 (t >> 2)
            const int a0 = (t & 3) + 8 * (t >> 2);
            const int f = acc[t];
            fmx = max(fmx, f);
            fmn = min(fmn, f);
            tcol[a0 * RS] = static_cast<int8_t>(f);
          };
          {
            // The tokens straight to global memory, in as few store instructions as the plain generator's (scattered stores
            // are paid per instruction): register group q holds the four consecutive rows 8 q + 4 h + (0..3); two
            // v_permlane32_swap hand the lower lane half rows 0..15 and the upper half rows 16..31 of the action's vector,
            // which leave as one 16-byte store per lane at S = 25 (store_vector16; 16 | 8 + 1 bytes = three instructions in round 3).
            if (TG_GF_ON(16)) {
              uint32_t X[4];
#pragma unroll
              for (int q = 0; q < 4; ++q)
                X[q] = pack4(acc[4 * q] + ga.shift, acc[4 * q + 1] + ga.shift, acc[4 * q + 2] + ga.shift, acc[4 * q + 3] + ga.shift);
              const auto s02 = __builtin_amdgcn_permlane32_swap(X[0], X[2], false, false);  // upper half's X0 <-> lower half's X2
              const auto s13 = __builtin_amdgcn_permlane32_swap(X[1], X[3], false, false);
              // lower half: s02 = (own X0: rows 0..3, the upper half's X0: rows 4..7), s13 = (own X1: 8..11, upper's X1: 12..15)
              // upper half: s02 = (the lower half's X2: rows 16..19, own X2: 20..23), s13 = (lower's X3: 24..27, own X3: 28..31)
              const int K[4] = {static_cast<int>(s02[0]), static_cast<int>(s02[1]), static_cast<int>(s13[0]), static_cast<int>(s13[1])};
              int8_t* const gp = ga.actions + g * blk + (3 * r + x) * S + 16 * h;
              struct __attribute__((packed)) P8 { uint32_t v[2]; };
              auto put = [&](int nb) {
                int o = 0;
                if (nb - o >= 16) {
                  const v4u_t qv{static_cast<uint32_t>(K[0]), static_cast<uint32_t>(K[1]), static_cast<uint32_t>(K[2]), static_cast<uint32_t>(K[3])};
                  asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(gp), "v"(qv) : "memory");
                  o = 16;
                }
                if (nb - o >= 8) { *reinterpret_cast<P8*>(gp + o) = P8{{static_cast<uint32_t>(K[o >> 2]), static_cast<uint32_t>(K[(o >> 2) + 1])}}; o += 8; }
                if (nb - o >= 4) { reinterpret_cast<UnalignedU32*>(gp + o)->v = static_cast<uint32_t>(K[o >> 2]); o += 4; }
                for (; o < nb; ++o) gp[o] = static_cast<int8_t>(static_cast<uint32_t>(K[o >> 2]) >> (8 * (o & 3)));
              };
              if constexpr (S > 16) {  // one 16-byte store per lane (store_vector16)
                const uint32_t Kv[4] = {static_cast<uint32_t>(K[0]), static_cast<uint32_t>(K[1]), static_cast<uint32_t>(K[2]), static_cast<uint32_t>(K[3])};
                store_vector16(Kv, ga.actions + g * blk + (3 * r + x) * S, true);
              } else if (h == 0) {
                put(S < 16 ? S : 16);
              }
            }
          }
#pragma unroll
          for (int t = 0; t < 16; ++t)
            if ((t & 3) + 8 * (t >> 2) + 4 < S) emit(t);  // valid in both lane halves
          if (h == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t)
              if ((t & 3) + 8 * (t >> 2) < S && (t & 3) + 8 * (t >> 2) + 4 >= S) emit(t);  // lower half only
          }
          big |= (fmx > lim) | (fmn < lim_lo);
          bad |= (fmx + ga.shift + 128) | (fmn + ga.shift + 128);  // a token outside int8 sets bits above the low byte
        }
      }
    }
  };

  // ---- LDS images of game g -> global memory, by threads t = 0 .. nthr-1: target (unless the exact path already wrote
  // it) as 16-byte chunks; tokens = the block's bytes [pad, pad + blk) of the image, aligned 16-byte chunks inside ----
  // `n` whole 16-byte chunks LDS -> global by threads t = 0 .. nthr-1, four chunks per thread and trip: the four LDS reads
  // are issued together, then the four stores (a read-wait-store chain per chunk made this phase, a few chunks per thread,
  // cost ~2 us of latency per game).  Reads past the end are clamped, not predicated; named values, not an array (hipcc
  // put a conditionally written uint4 array into scratch).
  auto copy_chunks = [&](const uint8_t* src, uint8_t* dst, int n, int t, int nthr) {
    for (int c0 = t; c0 < n; c0 += 4 * nthr) {
      const int c1 = c0 + nthr, c2 = c1 + nthr, c3 = c2 + nthr;
      const uint4 q0 = *reinterpret_cast<const uint4*>(src + 16 * c0);
      const uint4 q1 = *reinterpret_cast<const uint4*>(src + 16 * (c1 < n ? c1 : n - 1));
      const uint4 q2 = *reinterpret_cast<const uint4*>(src + 16 * (c2 < n ? c2 : n - 1));
      const uint4 q3 = *reinterpret_cast<const uint4*>(src + 16 * (c3 < n ? c3 : n - 1));
      // (unsigned 32-bit offsets from the wave-uniform base, each made opaque so that hipcc does not fold them into 64-bit
      // per-lane pointers: the stores take the scalar-base form)
      auto st16 = [&](uint32_t off, const uint4& q) {
        asm volatile("" : "+v"(off));
        *reinterpret_cast<uint4*>(dst + off) = q;
      };
      st16(16u * static_cast<uint32_t>(c0), q0);
      if (c1 < n) st16(16u * static_cast<uint32_t>(c1), q1);
      if (c2 < n) st16(16u * static_cast<uint32_t>(c2), q2);
      if (c3 < n) st16(16u * static_cast<uint32_t>(c3), q3);
    }
  };

  // ---- the target image of game g -> global memory as 16-byte chunks, by threads t = 0 .. nthr-1 (unless the exact path
  // already wrote it) ----
  auto store_outputs = [&](int64_t g, bool with_target, int t, int nthr) {
    if (with_target && TG_GF_ON(8)) {
      int8_t* const out = ga.target + g * ga.out_stride;
      constexpr int NFULL = G::TAIL ? G::NCHUNK - 1 : G::NCHUNK;  // whole 16-byte chunks
      copy_chunks(img, reinterpret_cast<uint8_t*>(out), NFULL, t, nthr);
      if (G::TAIL != 0 && t < G::TAIL) out[16 * NFULL + t] = static_cast<int8_t>(img[16 * NFULL + t]);
    }
  };

  // Workgroup OR of a per-thread flag with ONE barrier (HIP's __syncthreads_or costs three): non-zero flags are OR-ed
  // into an LDS word (rare), barrier, everyone reads the word.  Four words used round-robin; a word is cleared by
  // thread 0 after the NEXT barrier (every reader has passed it by then) and needed again three barriers later.
  uint32_t* const orw = reinterpret_cast<uint32_t*>(img + G::IMG + 32);
  if (tid < 4) orw[tid] = 0;
  int orslot = 0;
  auto wg_or = [&](int v) {
    if (v) atomicOr(&orw[orslot], static_cast<uint32_t>(v));
    lds_barrier();  // (not __syncthreads(): that would wait for the acknowledgement of every store of the previous game)
    const int r = static_cast<int>(orw[orslot]);
    if (tid == 0) orw[(orslot + 3) & 3] = 0;  // the word of the previous barrier
    orslot = (orslot + 1) & 3;
    return r;
  };

  __syncthreads();  // T and the OR words are zero
  int64_t prev = -1, cur = blockIdx.x;
  bool prev_exact = false;
  while (true) {
    const bool has_cur = cur < ga.B;
    // stores of the previous game: by everybody when nothing follows, else by the wavefronts that draw least
    if (prev >= 0) {
      if (!has_cur) store_outputs(prev, !prev_exact, tid, NTHREADS);
      else if (light) store_outputs(prev, !prev_exact, light_tid, light_threads);
    }
    if (!has_cur) break;
    int big = 0, bad = 0;  // big: factors beyond the byte products (exact fallback); bad: a token left int8 (flag)
    draw(cur, big, bad);
    const int verdict = wg_or((big ? 1 : 0) | ((bad & ~255) ? 2 : 0));  // B1: T, token image complete; image reads of prev done
    const bool exact = BASIS && (verdict & 1) != 0;  // workgroup-uniform; rare (drawn values are inside the byte products)
    int flag = 0;
    if (exact) {  // exact byte-wise form from the emitted tokens, straight to global memory
      note_fallback();
      __syncthreads();  // the tokens are in global memory: every wavefront's stores must have landed first (vmcnt + barrier)
      flag = exact_target_from_tokens<S, NTHREADS>(reinterpret_cast<const uint8_t*>(ga.actions + cur * blk), R, ga.shift,
                                                   ga.target + cur * ga.out_stride);
    } else {  // column tiles on the matrix cores -> the target image
      int hi = 0, lo = 0;
      if (TG_GF_ON(4)) accumulate_tiles<S, KS, NW, CHECK, LUT, (S % 4 != 0)>(T, img, Rp, tm, wave, col, h, hi, lo);
      flag = (hi > 127) | (lo < -128);
    }
    const bool any_ovf = (wg_or(flag) != 0) | ((verdict & 2) != 0);  // B2: the image is complete, T is free
    if (tid == 0 && any_ovf && ga.overflow) ga.overflow[cur] = 1;
    prev = cur;
    prev_exact = exact;
    cur += gridDim.x;
  }
}
