// tg_genfused.h -- the synthetic-demonstration generator in ONE kernel (included by tg_kernels.hip after tg_mfma.h).
//
// create_synthetic_demo (utils.py:203-233) / _create_synthetic_demos (datasets.py:124-142) for S = 9, 16, 25:
//   Philox -> factor bytes in registers -> [change of basis: one int8 MFMA per (mode, 32 actions)] -> the transposed
//   factor image T[x][r] and the token image in LDS -> the accumulation tiles of tg_mfma.h -> target + tokens out.
// Round 1 ran three kernels (gen_tokens_kernel, basis_tokens_mfma_kernel, genf_mfma_kernel): the tokens made a round
// trip through memory twice (19.7 MB written, transformed in place, re-read at S=25, R=64, B=4096) and the token
// kernel burnt a 32-bit Philox lane per 3-way draw.  Here the factors never leave the chip before they are final:
// per demo the kernel writes S^3 + 3SR bytes and reads nothing (with a basis: 3 S^2 bytes).
//
// Draw phase, one job = (mode x, 32 actions) per wavefront pass.  The lane mapping IS the int8 MFMA B-fragment
// mapping -- lane (col, h) owns elements k = 16h .. 16h+15 of the vector of action r0 + col -- so the change of basis
// D[a][r] = sum_i M_x[a][i] f_r[i] takes the drawn bytes as they stand (A fragment = row `col` of M_x, read from
// global memory), and without a basis the same registers go straight to T and to the token image.  One Philox block
// is eight 16-bit draws evaluated two at a time with packed int16 ops (draw_pair16); a lane runs the blocks of its
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
};

template <int S>
constexpr int genfused_lds_bytes(int Rp, int R) {
  return MGeo<S>::TROWS * (Rp + 16) + MGeo<S>::IMG + 32 + ((R * 3 * S + 15) & ~15) + 16 + 32;
}

// bytes 16h .. 16h+15 of an S-byte row in global memory (any alignment; nothing past the row is read; bytes >= S are 0)
template <int S>
__device__ __forceinline__ v4i row_fragment16(const int8_t* row, int h) {
  v4i f;
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

// The exact form of one game from its EMITTED tokens in LDS (any factor magnitude): the fallback of gen_fused_kernel
// for games whose transformed factors leave the byte-product range of the matrix-core path.  Whole workgroup.
template <int S>
__device__ __forceinline__ int exact_target_from_tokens(const uint8_t* tok, int R, int shift, int8_t* out) {
  constexpr int S2 = S * S, N = S2 * S, A3 = 3 * S;
  int ovf = 0;
  for (int e = threadIdx.x; e < N; e += kBlock) {
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
template <int S, int KS, bool BASIS>
__global__ __launch_bounds__(kBlock) void gen_fused_kernel(GenArgs ga, int Rp) {
  using G = MGeo<S>;
  extern __shared__ __attribute__((aligned(16))) uint8_t mfma_smem[];
  if constexpr (KS != 0) Rp = 32 * KS;
  const int RS = Rp + 16;
  const int R = ga.R, blk = R * G::A3;
  int8_t* const T = reinterpret_cast<int8_t*>(mfma_smem);
  uint8_t* const img = mfma_smem + G::TROWS * RS;
  uint8_t* const tokimg = img + G::IMG + 32;  // [pad + blk] bytes; pad = the block's 16-byte phase in global memory
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;

  // T starts all zero and only cells (x, i < S, r < R) are ever written: W rows l >= S and the padding actions
  // r >= R (up to Rp) stay zero for every game
  for (int e = 16 * tid; e < G::TROWS * RS; e += 16 * kBlock) *reinterpret_cast<uint4*>(T + e) = uint4{0, 0, 0, 0};

  TileMap<S> tm;
  make_tile_map<S>(tm, RS, wave, col, h);

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
  const int NTR = Rp >> 5, njob = 3 * NTR;
  __syncthreads();

  for (int64_t g = blockIdx.x; g < ga.B; g += gridDim.x) {
    const uint64_t gid = ga.gid0 + static_cast<uint64_t>(g);
    int8_t* const gtok = ga.actions + g * blk;
    const int pad = static_cast<int>(reinterpret_cast<uintptr_t>(gtok) & 15);
    uint8_t* const tk = tokimg + pad;
    int bad = 0, big = 0;  // bad: a token left int8 (flag); big: factors beyond the byte products (exact fallback)

    // ---- 1. draw (and transform) the factors: registers -> T and token image ----
    for (int job = wave; job < njob; job += kBlock / 64) {
      const int x = job / NTR, r = 32 * (job - x * NTR) + col;
      const bool active = r < R;
      v4i fa;
      if constexpr (BASIS)  // row a = col of M_x (rows >= S shadow the last one; their results are never used)
        fa = row_fragment16<S>(ga.basis + ((g * 3 + x) * S + (col < S ? col : S - 1)) * S, h);
      uint32_t Dw[4] = {0, 0, 0, 0};  // factor bytes of this lane's blocks (two dwords per block)
      uint32_t Kw[4] = {0, 0, 0, 0};  // the same as tokens (value + shift)
      bool need = active && draws;
      uint32_t attempt = 0;
      while (true) {
        if (need) {
#pragma unroll
          for (int b = 0; b < (kSwap ? 1 : 2); ++b) {
            if (8 * (q0 + b) < S) {
              const U4 o = philox4x32_10(U4{static_cast<uint32_t>(gid), static_cast<uint32_t>(gid >> 32),
                                            static_cast<uint32_t>(3 * r + x), (attempt << 8) | static_cast<uint32_t>(q0 + b)},
                                         k0, k1);
              const uint32_t P0 = draw_pair16(o.x, ga.D), P1 = draw_pair16(o.y, ga.D);
              const uint32_t P2 = draw_pair16(o.z, ga.D), P3 = draw_pair16(o.w, ga.D);
              Dw[2 * b] = __builtin_amdgcn_perm(P1, P0, 0x06040200u) & vmask[2 * b];
              Dw[2 * b + 1] = __builtin_amdgcn_perm(P3, P2, 0x06040200u) & vmask[2 * b + 1];
              if constexpr (!BASIS) {
                Kw[2 * b] = __builtin_amdgcn_perm(pk_add_u16(P1, shp), pk_add_u16(P0, shp), 0x06040200u);
                Kw[2 * b + 1] = __builtin_amdgcn_perm(pk_add_u16(P3, shp), pk_add_u16(P2, shp), 0x06040200u);
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
        if (active && holds) {
          int8_t* const tcol = T + (x * S + kbase) * RS + r;
          uint8_t* const trow = tk + (3 * r + x) * S + kbase;
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            if (k < S && kbase + k < S) {
              tcol[k * RS] = static_cast<int8_t>(static_cast<uint32_t>(F[k >> 2]) >> (8 * (k & 3)));
              trow[k] = static_cast<uint8_t>(static_cast<uint32_t>(K[k >> 2]) >> (8 * (k & 3)));
            }
          }
        }
      } else {
        // D[a][r] = sum_i M_x[a][i] f_r[i]: one int8 MFMA; this lane gets a = (t & 3) + 8 (t >> 2) + 4 h of action r
        v16i acc;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0;
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, F, acc, 0, 0, 0);
        if (active) {
          int8_t* const tcol = T + (x * S) * RS + r;
          uint8_t* const trow = tk + (3 * r + x) * S;
          const int lim = x < 2 ? G::UVLIM : 127;
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const int a = (t & 3) + 8 * (t >> 2) + 4 * h;
            if ((t & 3) + 8 * (t >> 2) < S && a < S) {
              const int f = acc[t], tokv = f + ga.shift;
              big |= (f > lim) | (f < -lim - (x < 2 ? 0 : 1));
              bad |= tokv + 128;
              tcol[a * RS] = static_cast<int8_t>(f);
              trow[a] = static_cast<uint8_t>(tokv);
            }
          }
        }
      }
    }
    const int verdict = __syncthreads_or((big ? 1 : 0) | ((bad & ~255) ? 2 : 0));  // also: T and the token image are complete
    int8_t* const out = ga.target + g * ga.out_stride;
    bool any_ovf = (verdict & 2) != 0;
    if (verdict & 1) {  // workgroup-uniform; rare: exact byte-wise form from the emitted tokens
      note_fallback();
      any_ovf |= __syncthreads_or(exact_target_from_tokens<S>(tk, R, ga.shift, out)) != 0;
    } else {
      // ---- 2. column tiles on the matrix cores ----
      int hi = 0, lo = 0;
      accumulate_tiles<S, KS>(T, img, Rp, tm, wave, col, h, hi, lo);
      any_ovf |= __syncthreads_or((hi > 127) | (lo < -128)) != 0;  // also: the image is complete
      // ---- 3. image -> global, 16-byte chunks ----
      for (int c = tid; c < G::NCHUNK; c += kBlock)
        store_chunk<G::TAIL>(out + 16 * c, *reinterpret_cast<const uint4*>(img + 16 * c), c == G::NCHUNK - 1);
    }
    // ---- 4. tokens -> global: the block's bytes [pad, pad + blk) of the image, aligned 16-byte chunks inside ----
    {
      uint8_t* const gbase = reinterpret_cast<uint8_t*>(gtok) - pad;
      const int total = pad + blk, nchunk = (total + 15) >> 4;
      for (int c = tid; c < nchunk; c += kBlock) {
        const int b0 = 16 * c, b1 = b0 + 16;
        if (b0 >= pad && b1 <= total) {
          *reinterpret_cast<uint4*>(gbase + b0) = *reinterpret_cast<const uint4*>(tokimg + b0);
        } else {
          for (int b = b0 < pad ? pad : b0; b < (b1 < total ? b1 : total); ++b) gbase[b] = tokimg[b];
        }
      }
    }
    if (tid == 0 && any_ovf && ga.overflow) ga.overflow[g] = 1;
    __syncthreads();  // T and the images are reused by the next game
  }
}
