// tg_mfma.h -- accumulation over many rank-1 terms on the matrix cores (included by tg_kernels.hip):
// genf_mfma_kernel (tg_gen_from_factors_i8, the generator) and many_mfma_kernel (tg_step_many_i8, below).
//
// target[i][j][l] = sum_r u_r[i] v_r[j] w_r[l]   (reference utils.py:218-232, datasets.py:127-141)
//
// is the dense contraction on the path: per game a (32 x R) by (R x S^2) integer product,
//     D[l][(i,j)] = sum_r  W[l][r] * P[r][(i,j)],      P[r][(i,j)] = u_r[i] * v_r[j],
// with R ops per output byte (R = 64 at BASELINE config 5) -- compute-bound on the vector ALU
// (tg_rows.h: 117 us for 4096 games at S = 25, R = 64, 44 % of the v_pk_mad_i16 rate), a small
// fraction of the int8 MFMA rate.  One workgroup per game at a time (several games per workgroup):
//   1. the game's R x 3S factors (token - shift) are written to LDS TRANSPOSED, T[x][r], so that
//      the 16 consecutive r a lane needs are one ds_read_b128 (tokens arrive by buffer loads: one
//      offset register, rows past R read as zero);
//   2. a wavefront owns column tiles of 32 columns n = (i,j).  Per 32 values of r: the A fragment is
//      W (row l = lane & 31), the B fragment is built by the lane for its column: 16 byte products
//      u_r[i] * v_r[j] (v_mul_i32_i24 with SDWA byte selects, written straight into the bytes of the
//      fragment), then one v_mfma_i32_32x32x32_i8.  Products must fit int8: |u|, |v| <= 11
//      (checked while staging; a game beyond that is done by slow_game).  Sums are int32: exact.
//   3. "left int8" is a running max / min over the int32 results (v_max3 / v_min3).  The low bytes
//      go into a dense S^3 image in LDS.  For odd S a column (S bytes) is not dword aligned and
//      gfx950's LDS stalls on unaligned dwords (SQ_LDS_UNALIGNED_STALL was half of the kernel), so
//      two tiles are finished together, v_permlane32_swap gives every lane one WHOLE column, and the
//      lane writes the aligned dwords that start inside it (v_alignbyte; the dword that closes the
//      column takes its last bytes from the next lane through DPP);
//   4. the image goes out as aligned 16-byte chunks.
// Built with -mllvm -amdgpu-mfma-vgpr-form: the results land in VGPRs, no v_accvgpr moves.
// A and B fragments use the same lane -> r mapping, so the result does not depend on how the
// instruction orders k inside a fragment; C/D is the 32x32 map of the guide (col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).
#pragma once

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int S>
struct MGeo {
  static constexpr int N = S * S * S, S2 = S * S, A3 = 3 * S;
  static constexpr int NT = (S2 + 31) / 32;   // column tiles per game
  static constexpr int TROWS = 2 * S + 32;    // rows of T: u (S), v (S), w padded to 32
  static constexpr int NCHUNK = (N + 15) / 16;
  static constexpr int TAIL = N % 16;
  static constexpr int IMG = NCHUNK * 16;     // bytes of the output image
  static constexpr int UVLIM = 11;            // |u|, |v| <= 11: u * v fits int8
};

// bytes of dynamic LDS for R actions padded to Rp (a multiple of 32)
template <int S>
constexpr int mfma_lds_bytes(int Rp) { return MGeo<S>::TROWS * (Rp + 16) + MGeo<S>::IMG + 32; }

// 16 byte-wise products a[k] * b[k] (signed) -> bytes of the result.  The four dwords are
// independent chains interleaved so that an instruction never reads the register the previous one
// wrote with a byte dst_sel; the closing s_nop covers "VALU write -> MFMA operand".
__device__ __forceinline__ v4i bytemul16(const v4i a, const v4i b) {
  v4i d;
#define TG_BM(D, A, B, SEL, KEEP)                                                                     \
  "v_mul_i32_i24_sdwa " D ", sext(" A "), sext(" B ") dst_sel:" SEL " dst_unused:" KEEP " src0_sel:" SEL \
  " src1_sel:" SEL "\n\t"
  asm(TG_BM("%0", "%4", "%8", "BYTE_0", "UNUSED_PAD") TG_BM("%1", "%5", "%9", "BYTE_0", "UNUSED_PAD")
      TG_BM("%2", "%6", "%10", "BYTE_0", "UNUSED_PAD") TG_BM("%3", "%7", "%11", "BYTE_0", "UNUSED_PAD")
      TG_BM("%0", "%4", "%8", "BYTE_1", "UNUSED_PRESERVE") TG_BM("%1", "%5", "%9", "BYTE_1", "UNUSED_PRESERVE")
      TG_BM("%2", "%6", "%10", "BYTE_1", "UNUSED_PRESERVE") TG_BM("%3", "%7", "%11", "BYTE_1", "UNUSED_PRESERVE")
      TG_BM("%0", "%4", "%8", "BYTE_2", "UNUSED_PRESERVE") TG_BM("%1", "%5", "%9", "BYTE_2", "UNUSED_PRESERVE")
      TG_BM("%2", "%6", "%10", "BYTE_2", "UNUSED_PRESERVE") TG_BM("%3", "%7", "%11", "BYTE_2", "UNUSED_PRESERVE")
      TG_BM("%0", "%4", "%8", "BYTE_3", "UNUSED_PRESERVE") TG_BM("%1", "%5", "%9", "BYTE_3", "UNUSED_PRESERVE")
      TG_BM("%2", "%6", "%10", "BYTE_3", "UNUSED_PRESERVE") TG_BM("%3", "%7", "%11", "BYTE_3", "UNUSED_PRESERVE")
      "s_nop 1"
      : "=&v"(d.x), "=&v"(d.y), "=&v"(d.z), "=&v"(d.w)
      : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
#undef TG_BM
  return d;
}

// The same sixteen products for the ternary vocabulary {-1, 0, 1} by table lookup (gen_fused_kernel<.., LUT>): the u
// rows of T hold the code u + 1 (0, 1, 2), the v rows the code 4 m(v) with m(-1) = 0, m(+1) = 1, m(0) = 2, so
// code_u | code_v is a v_perm_b32 selector into an 8-byte pool: selectors 0..2 -> (+1, 0, -1) = u * (-1), 4..6 ->
// (-1, 0, +1) = u * (+1), and 8..10 (v = 0) replicate the sign bits of pool bytes 1, 3, 5, which are zero.  One v_or_b32
// (VOP2) + one v_perm_b32 per four products instead of four SDWA multiplies.
__device__ __forceinline__ v4i lutmul16(const v4i a, const v4i b, uint32_t pool_hi, uint32_t pool_lo) {
  v4i d;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    d[q] = static_cast<int>(__builtin_amdgcn_perm(pool_hi, pool_lo, static_cast<uint32_t>(a[q]) | static_cast<uint32_t>(b[q])));
  return d;
}
constexpr uint32_t kLutPoolLo = 0x00FF0001u, kLutPoolHi = 0x000100FFu;  // bytes 0..3 = (+1, 0, -1, 0), 4..7 = (-1, 0, +1, 0)
constexpr uint32_t kLutCodeV = 0x00040800u;                             // v code by u code: bytes (0, 8, 4, 0)

typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
struct __attribute__((packed)) UnalignedU32 { uint32_t v; };  // gfx950 LDS takes unaligned dwords (ds_write_b32)

// ---- the tile phase shared by genf_mfma_kernel and gen_fused_kernel (tg_genfused.h) -----------------------------
// Which column tiles a wavefront owns, and where its lanes read their fragments: the same for every game.
template <int S, int NW_ = kBlock / 64>  // NW_ wavefronts share the NT column tiles of a game
struct TileMap {
  static constexpr int NW = NW_, TPW = (MGeo<S>::NT + NW - 1) / NW;
  int uoff[TPW], voff[TPW], ncol[TPW];
  int woff;
};

template <int S, int NW_>
__device__ __forceinline__ void make_tile_map(TileMap<S, NW_>& tm, int RS, int wave, int col, int h) {
  using G = MGeo<S>;
  constexpr int NW = NW_, TPW = TileMap<S, NW_>::TPW;
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    const int n = 32 * (wave + NW * k) + col;
    const int nn = n < G::S2 ? n : G::S2 - 1;  // columns past S^2 shadow the last one; never stored
    const int i = nn / S, j = nn - i * S;
    tm.uoff[k] = i * RS + 16 * h;
    tm.voff[k] = (S + j) * RS + 16 * h;
    tm.ncol[k] = n < G::S2 ? n : -1;
  }
  tm.woff = (2 * S + col) * RS + 16 * h;
}

// All column tiles of one game: T (transposed factors, rows of RS = Rp + 16 bytes) -> the dense S^3 byte image `img`.
// hi / lo: running max / min of every int32 result of this lane (the int8 range check).
// CHECK = false: the caller has proved that no result can leave int8 (R * fmax^3 <= 127): the running max / min
// (sixteen VALU instructions per tile) is dropped and hi / lo stay untouched.
// LUT = true: T's u / v rows hold the ternary codes of lutmul16 instead of the factors (w rows: the factors, as ever).
// BYTES = true (round 3): every result leaves as ONE ds_write_b8 straight from its accumulator register (the low byte of
// the int32 IS the int8 result; a byte store has no alignment to respect): no packing (12 v_perm_b32 per tile), no
// half-wave exchange, no v_alignbyte -- ~20 VALU instructions per tile move to the LDS pipe, which has the room (the
// kernels are bound by VALU issue: SQ_ACTIVE_INST_LDS is a fifth of SQ_ACTIVE_INST_VALU).
template <int S, int KS, int NW_, bool CHECK = true, bool LUT = false, bool BYTES = false>
__device__ __forceinline__ void accumulate_tiles(const int8_t* T, uint8_t* img, int Rp, const TileMap<S, NW_>& tm, int wave,
                                                 int col, int h, int& hi, int& lo) {
  using G = MGeo<S>;
  constexpr int NW = NW_, TPW = TileMap<S, NW_>::TPW;
  const int (&uoff)[TPW] = tm.uoff;
  const int (&voff)[TPW] = tm.voff;
  const int (&ncol)[TPW] = tm.ncol;
  const int woff = tm.woff;
    v4i wa[KS ? KS : 1];
#pragma unroll
    for (int k = 0; k < KS; ++k) wa[k] = *reinterpret_cast<const v4i*>(T + woff + 32 * k);
    uint32_t pool_hi = kLutPoolHi, pool_lo = kLutPoolLo;  // (VOP3 on gfx950 takes one scalar operand and no literal)
    if constexpr (LUT) asm volatile("" : "+v"(pool_hi), "+s"(pool_lo));
    auto products = [&](const v4i a, const v4i b) {
      if constexpr (LUT) return lutmul16(a, b, pool_hi, pool_lo);
      else return bytemul16(a, b);
    };
    // the accumulators of one tile: acc[t] = row l = (t & 3) + 8 (t >> 2) + 4 h of this lane's column
    auto tile_acc = [&](int k) {
      v16i acc;
#pragma unroll
      for (int t2 = 0; t2 < 16; ++t2) acc[t2] = 0;
      if constexpr (KS != 0) {
        v4i p[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          p[ks] = products(*reinterpret_cast<const v4i*>(T + uoff[k] + 32 * ks),
                           *reinterpret_cast<const v4i*>(T + voff[k] + 32 * ks));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa[ks], p[ks], acc, 0, 0, 0);
      } else {
        for (int k0 = 0; k0 < Rp; k0 += 32) {
          const v4i w = *reinterpret_cast<const v4i*>(T + woff + k0);
          const v4i p = products(*reinterpret_cast<const v4i*>(T + uoff[k] + k0),
                                 *reinterpret_cast<const v4i*>(T + voff[k] + k0));
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, p, acc, 0, 0, 0);
        }
      }
      if constexpr (CHECK) {
#pragma unroll
        for (int t2 = 0; t2 < 16; t2 += 2) {
          hi = max(max(acc[t2], acc[t2 + 1]), hi);
          lo = min(min(acc[t2], acc[t2 + 1]), lo);
        }
      }
      return acc;
    };
    if constexpr (BYTES) {
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        if (G::NT % NW != 0 && wave + NW * k >= G::NT) break;  // wave-uniform
        __builtin_amdgcn_sched_barrier(0);
        const v16i acc = tile_acc(k);
        if constexpr (CHECK) asm volatile("" : "+v"(hi), "+v"(lo));  // the range of THIS tile is final: one tile live at a time
        if (ncol[k] >= 0) {
          // volatile: hipcc otherwise merges neighbouring byte stores back into (unaligned) dwords, packing included
          volatile lds_u8_t* const dst = (lds_u8_t*)(img + ncol[k] * S + 4 * h);
#pragma unroll
          for (int t2 = 0; t2 < 16; ++t2)  // rows valid in both lane halves
            if ((t2 & 3) + 8 * (t2 >> 2) + 4 < S) dst[(t2 & 3) + 8 * (t2 >> 2)] = static_cast<uint8_t>(acc[t2]);
          if (h == 0) {
#pragma unroll
            for (int t2 = 0; t2 < 16; ++t2)  // rows of the lower half only
              if ((t2 & 3) + 8 * (t2 >> 2) < S && (t2 & 3) + 8 * (t2 >> 2) + 4 >= S)
                dst[(t2 & 3) + 8 * (t2 >> 2)] = static_cast<uint8_t>(acc[t2]);
          }
        }
      }
      return;
    }
    // one tile -> X[q] = the low bytes of this lane's results for rows l = 8 q + 4 h + (0..3)
    auto tile = [&](int k, uint32_t (&X)[4]) {
      v16i acc;
#pragma unroll
      for (int t2 = 0; t2 < 16; ++t2) acc[t2] = 0;
      if constexpr (KS != 0) {
        v4i p[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          p[ks] = products(*reinterpret_cast<const v4i*>(T + uoff[k] + 32 * ks),
                           *reinterpret_cast<const v4i*>(T + voff[k] + 32 * ks));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa[ks], p[ks], acc, 0, 0, 0);
      } else {
        for (int k0 = 0; k0 < Rp; k0 += 32) {
          const v4i w = *reinterpret_cast<const v4i*>(T + woff + k0);
          const v4i p = products(*reinterpret_cast<const v4i*>(T + uoff[k] + k0),
                                 *reinterpret_cast<const v4i*>(T + voff[k] + k0));
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, p, acc, 0, 0, 0);
        }
      }
      if constexpr (CHECK) {
#pragma unroll
        for (int t2 = 0; t2 < 16; t2 += 2) {
          hi = max(max(acc[t2], acc[t2 + 1]), hi);
          lo = min(min(acc[t2], acc[t2 + 1]), lo);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) X[q] = pack4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
      asm volatile("" : "+v"(hi), "+v"(lo), "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]));  // one tile live at a time
    };
#pragma unroll
    for (int k = 0; k < TPW; k += 2) {
      if (G::NT % NW != 0 && wave + NW * k >= G::NT) break;  // wave-uniform
      __builtin_amdgcn_sched_barrier(0);
      uint32_t XA[4], XB[4] = {0, 0, 0, 0};
      tile(k, XA);
      const bool pair = k + 1 < TPW && (G::NT % NW == 0 || wave + NW * (k + 1) < G::NT);  // wave-uniform
      if (pair) tile(k + 1, XB);
      // Lane (col, h) of tile A holds rows 8q+4h+(0..3) of column col; its partner lane (col, 1-h) holds the
      // others.  v_permlane32_swap exchanges XA of the upper half-wave with XB of the lower one: afterwards the
      // LOWER half-wave owns whole columns of tile A, the UPPER half-wave whole columns of tile B:
      // U[q] = rows 8q..8q+3, V[q] = rows 8q+4..8q+7 of the lane's column.
      uint32_t E[9];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const auto sw = __builtin_amdgcn_permlane32_swap(XA[q], XB[q], false, false);
        E[2 * q] = sw[0];
        E[2 * q + 1] = sw[1];
      }
      E[8] = 0;
      const int nown = h ? (pair ? ncol[k + 1 < TPW ? k + 1 : k] : -1) : ncol[k];
      // The column is S bytes at image offset S*n: not dword aligned for odd S, and gfx950's LDS stalls on
      // unaligned dwords (SQ_LDS_UNALIGNED_STALL was half of the kernel).  So every lane writes the ALIGNED
      // dwords that START inside its column; the last of them is completed with the first bytes of the next
      // column, fetched from the next lane (a tile's 32 columns are 32 S bytes: 16-byte aligned at both ends).
      constexpr int Q = (S + 3) / 4, r = S % 4;
      static_assert(r <= 1, "tg_mfma.h: column emission handles S % 4 in {0, 1}");
      if constexpr (r == 1) {
        const uint32_t nxt = __builtin_amdgcn_update_dpp(0u, E[0], 0x130, 0xf, 0xf, false);  // lane + 1's first dword
        E[Q - 1] |= nxt << 8;  // rows >= S of D are zero (the W rows there are), so the upper bytes were 0
      }
      if (nown >= 0) {
        const int base = nown * S;
        const int o = (4 - (base & 3)) & 3;  // first aligned dword starting inside the column
        uint8_t* dst = img + base + o;
#pragma unroll
        for (int jd = 0; jd < Q; ++jd) {
          if (4 * jd + o < S) {  // o == 0 for every lane when S % 4 == 0
            const uint32_t d = __builtin_amdgcn_alignbyte(E[jd + 1], E[jd], static_cast<uint32_t>(o));
            *reinterpret_cast<uint32_t*>(dst + 4 * jd) = d;
          }
        }
      }
    }
}

// KS: k-steps of 32 actions known at compile time (1 or 2: R <= 64), 0 = run-time Rp / 32.
template <int S, int KS>
__global__ __launch_bounds__(kBlock) void genf_mfma_kernel(ApplyArgs a, int Rp) {
  using G = MGeo<S>;
  extern __shared__ __attribute__((aligned(16))) uint8_t mfma_smem[];
  if constexpr (KS != 0) Rp = 32 * KS;
  const int RS = Rp + 16;  // row stride of T: a multiple of 16 (ds_read_b128), 5 bank quads apart at Rp = 64
  int8_t* const T = reinterpret_cast<int8_t*>(mfma_smem);
  uint8_t* const img = mfma_smem + G::TROWS * RS;
  uint8_t* const flags = img + G::IMG + 16;  // the last column's closing dword may spill <= 3 bytes past N
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int R = a.nact;

  // rows l >= S of W never receive a factor: zero them once, so that the unused rows of D are zero
  for (int e = tid; e < (32 - S) * RS; e += kBlock) T[(3 * S) * RS + e] = 0;

  // staging role: lane = (token position x, one of NRG groups of actions); each trip handles 4 actions
  constexpr int NRG = kBlock / G::A3;
  constexpr int TB = 6;  // trips per batch: every byte load of a batch is in flight before one is used
  const int sx = tid % G::A3, srg = tid / G::A3;
  const int slo = sx < 2 * S ? -G::UVLIM : -128, shi = sx < 2 * S ? G::UVLIM : 127;

  // tile role: this wavefront's column tiles are the same for every game
  TileMap<S> tm;
  make_tile_map(tm, RS, wave, col, h);

  // Token fetch and staging.  Buffer loads: one VGPR of lane offset + immediates instead of 24 64-bit addresses,
  // and rows r >= R fall outside num_records (they read as 0), so no clamping.  fetch() only issues the loads of
  // one batch (TB trips of 4 actions); commit() turns a batch into factors, checks their range and writes T.
  auto fetch = [&](int64_t gg, int rb, int (&f)[TB][4]) {
    const __amdgpu_buffer_rsrc_t tok = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int8_t*>(a.actions + gg * R * G::A3), 0, R * G::A3, 0x00027000);
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = rb + 4 * NRG * tb + t;
        f[tb][t] = static_cast<int8_t>(__builtin_amdgcn_raw_buffer_load_b8(tok, sx + r * G::A3, 0, 0));
      }
  };
  auto commit = [&](int rb, int (&f)[TB][4], int& big) {
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
      const int r0 = rb + 4 * NRG * tb;
      if (r0 < Rp) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f[tb][t] = (r0 + t < R) ? f[tb][t] - a.shift : 0;
          big |= (f[tb][t] < slo) | (f[tb][t] > shi);
        }
        *reinterpret_cast<uint32_t*>(T + sx * RS + r0) = pack4(f[tb][0], f[tb][1], f[tb][2], f[tb][3]);
      }
    }
  };
  // When one batch covers all the actions (R <= 4 NRG TB), the NEXT game's tokens are fetched right after this
  // game's staging barrier and stay in registers through the tile phase: the memory latency (about a quarter of
  // a game's time in this kernel) disappears behind the arithmetic.
  const bool one_batch = Rp <= 4 * NRG * TB;  // uniform
  int pf[TB][4];
  if (one_batch && srg < NRG && static_cast<int64_t>(blockIdx.x) < a.B) fetch(blockIdx.x, 4 * srg, pf);

  for (int64_t g = blockIdx.x; g < a.B; g += gridDim.x) {
    // ---- 1. factors of this game, transposed into LDS (4 actions = one dword); range check; r >= R -> 0 ----
    int big = 0;
    if (srg < NRG) {
      if (one_batch) {
        commit(4 * srg, pf, big);
      } else {
        for (int rb = 4 * srg; rb < Rp; rb += 4 * NRG * TB) {
          int f[TB][4];
          fetch(g, rb, f);
          commit(rb, f, big);
        }
      }
    }
    const bool too_big = __syncthreads_or(big);
    if (one_batch && srg < NRG && g + gridDim.x < a.B) fetch(g + gridDim.x, 4 * srg, pf);
    if (too_big) {  // workgroup-uniform; rare: exact byte-wise form
      note_fallback();
      slow_game<GENF>(a, g, flags);
      __syncthreads();
      continue;
    }

    // ---- 2. column tiles on the matrix cores, two at a time ----
    int hi = 0, lo = 0;  // running max / min of every result of this lane
    accumulate_tiles<S, KS, kBlock / 64, true, false, (S % 4 != 0)>(T, img, Rp, tm, wave, col, h, hi, lo);
    const bool any_ovf = __syncthreads_or((hi > 127) | (lo < -128));  // also: the image is complete

    // ---- 3. image -> global, 16-byte chunks ----
    int8_t* out = a.out + g * a.out_stride;
    for (int c = tid; c < G::NCHUNK; c += kBlock)
      store_chunk<G::TAIL>(out + 16 * c, *reinterpret_cast<const uint4*>(img + 16 * c), c == G::NCHUNK - 1);
    if (tid == 0 && any_ovf && a.overflow) a.overflow[g] = 1;
    __syncthreads();  // T and the image are reused by the next game
  }
}

// =============================================================================================
// tg_step_many_i8 on the matrix cores: out = in - sum_k A_k, done_step, overflow.
//
// The final state is one more accumulation (the input enters through an identity fragment:
// D[l][n] += sum_l' I[l][l'] X0[n][l'] puts X0 into the accumulators in the result's own layout, so it
// costs one MFMA per tile and no unpacking).  What a sum cannot give directly is what happened BETWEEN
// the K steps; both questions are answered exactly, with the fast path as a filter:
//   * overflow: every prefix P_k = P_K + sum_{r>k} A_r, so |P_k| <= max|P_K| + sum_r mu_r mv_r mw_r
//     (m?_r = largest |factor| of action r).  If that bound is <= 127 no step can have left int8.
//   * done_step: two linear functionals h(X) = sum c[i,j,l] X[i,j,l] mod 2^32 with rank-1 weights
//     c = pu (x) pv (x) pw, so h(A_r) = (pu.u_r)(pv.v_r)(pw.w_r) costs 3S MACs per action, and h(X0) comes
//     out of two spare rows of the identity fragment (pw) times a per-column weight (pu[i] pv[j]).
//     P_k = 0 implies h(P_k) = 0, so the first zero state is the first k where both functionals vanish --
//     or a false positive (probability ~2^-64).  A candidate at k = K-1 is checked against the final
//     state, which is already there.
//     That scalar bound costs nothing but is coarse (K <= 127 for {-1,0,1}; the paper's {-2..2} fails it at once).
//     Games that fail it take the ELEMENTWISE bound |P_k[e]| <= |X0[e]| + sum_r |u_r[i]| |v_r[j]| |w_r[l]|: one more
//     pass over the tiles, the same accumulation on the absolute values (|T| copied next to T, |X0| through an
//     identity fragment), only the maximum is kept.  If it is <= 127 everywhere no prefix left int8.
// Games that fail both bounds, have a candidate before K-1, or have factors beyond the byte-product range
// are NOT written: their done_step is set to kNeedsExact and the lattice kernels (tg_rows.h / tg_packed.h),
// launched right after with ApplyArgs::only_flagged, redo exactly those games (counted in g_many_handovers).
// =============================================================================================

struct FunctionalWeights {
  int uv[2][2][32];  // [functional][u or v][index]: odd, 11 bits, signed: pu*pv fits 22 bits (24-bit multiplies)
  int w[2][32];      // [functional][index]: odd, int8 (they ride in the identity fragment)
};
constexpr FunctionalWeights make_functional_weights() {
  FunctionalWeights t{};
  uint32_t s = 0x9E3779B9u;
  for (int m = 0; m < 2; ++m) {
    for (int x = 0; x < 2; ++x)
      for (int i = 0; i < 32; ++i) {
        s = s * 1664525u + 1013904223u;
        t.uv[m][x][i] = static_cast<int>((s >> 21) | 1u) - (1 << 10);
      }
    for (int i = 0; i < 32; ++i) {
      s = s * 1664525u + 1013904223u;
      t.w[m][i] = static_cast<int>((s >> 24) | 1u) - 128;
    }
  }
  return t;
}
__constant__ FunctionalWeights g_fw = make_functional_weights();

template <int S>
constexpr int many_mfma_lds_bytes(int Rp) {
  return 2 * MGeo<S>::TROWS * (Rp + 16) + MGeo<S>::IMG + 32 + 32 + 8 * MGeo<S>::NT * 32 + 36 * Rp;
}

// |x| of four packed int8 (-128 stays 0x80: callers exclude it)
__device__ __forceinline__ uint32_t abs4_i8(uint32_t x) {
  const uint32_t s1 = (x >> 7) & 0x01010101u;
  return (x ^ (s1 * 0xFFu)) + s1;
}

template <int S, int KS>
__global__ __launch_bounds__(kBlock, 4) void many_mfma_kernel(ApplyArgs a, int Rp) {
  using G = MGeo<S>;
  static_assert((S % 8) < 3, "the two functional rows S, S+1 must be rows of the lower half-wave");
  extern __shared__ __attribute__((aligned(16))) uint8_t mfma_smem[];
  if constexpr (KS != 0) Rp = 32 * KS;
  const int RS = Rp + 16;
  int8_t* const T = reinterpret_cast<int8_t*>(mfma_smem);
  uint8_t* const img = mfma_smem + G::TROWS * RS;
  int* const red = reinterpret_cast<int*>(img + G::IMG + 32);  // [0,1] h(X0), [2] max |final|
  uint32_t* const cw = reinterpret_cast<uint32_t*>(red + 8);    // [2][NT*32] column weights pu[i] pv[j]
  int* const sdot = reinterpret_cast<int*>(cw + 2 * G::NT * 32);  // [2][3][Rp]
  int* const smx = sdot + 6 * Rp;                                 // [3][Rp]
  int8_t* const Tabs = reinterpret_cast<int8_t*>(smx + 3 * Rp);   // |T|, same layout (filled for games on the elementwise bound)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int R = a.nact;

  for (int e = tid; e < (32 - S) * RS; e += kBlock) T[(3 * S) * RS + e] = 0;
  // The functional weights come to LDS first (one coalesced load + a barrier): per-lane lookups in the __constant__
  // table are vector loads, and the set-up chained four of their latencies (3.9 us per workgroup of four games).
  int* const wl = reinterpret_cast<int*>(Tabs);  // set-up only: [2][2][32] uv weights, then [2][32] w weights
  constexpr int kWL = sizeof(FunctionalWeights) / sizeof(int);
  static_assert(kWL == 192 && kWL <= kBlock, "one weight per thread");
  if (tid < kWL) wl[tid] = (&g_fw.uv[0][0][0])[tid];
  __syncthreads();
  auto wuv = [&](int m, int x, int i) { return wl[(m * 2 + x) * 32 + i]; };
  auto ww = [&](int m, int i) { return wl[128 + m * 32 + i]; };
  for (int n = tid; n < G::NT * 32; n += kBlock) {
    const int nn = n < G::S2 ? n : G::S2 - 1;
    const int i = nn / S, j = nn - i * S;
#pragma unroll
    for (int m = 0; m < 2; ++m) cw[m * G::NT * 32 + n] = n < G::S2 ? static_cast<uint32_t>(wuv(m, 0, i) * wuv(m, 1, j)) : 0u;
  }

  constexpr int NRG = kBlock / G::A3;
  constexpr int TB = 6;
  const int sx = tid % G::A3, srg = tid / G::A3;
  const int slo = sx < 2 * S ? -G::UVLIM : -127, shi = sx < 2 * S ? G::UVLIM : 127;  // (w = -128 has no int8 |w|)

  constexpr int NW = kBlock / 64, TPW = (G::NT + NW - 1) / NW;
  int uoff[TPW], voff[TPW], ncol[TPW], xoff[TPW];
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    const int n = 32 * (wave + NW * k) + col;
    const int nn = n < G::S2 ? n : G::S2 - 1;
    const int i = nn / S, j = nn - i * S;
    uoff[k] = i * RS + 16 * h;
    voff[k] = (S + j) * RS + 16 * h;
    ncol[k] = n < G::S2 ? n : -1;
    xoff[k] = nn * S + 16 * h;  // this lane's 16 bytes of X0's column nn (bytes past the column meet zero rows)
  }
  const int woff = (2 * S + col) * RS + 16 * h;
  // identity fragment with the two functional rows: row l = col, k = 16 h + j
  v4i ida;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t w = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = 16 * h + 4 * d + t;
      int v = 0;
      if (k < S) v = col < S ? (k == col) : (col == S ? ww(0, k) : (col == S + 1 ? ww(1, k) : 0));
      w |= static_cast<uint32_t>(v & 255) << (8 * t);
    }
    ida[d] = static_cast<int>(w);
  }
  constexpr int kF0 = (S & 3) + 4 * (S >> 3), kF1 = ((S + 1) & 3) + 4 * ((S + 1) >> 3);  // registers of rows S, S+1 (h = 0)
  v4i idp;  // the plain identity (rows >= S zero): |X0| enters the bound pass through it
#pragma unroll
  for (int d = 0; d < 4; ++d) idp[d] = col < S ? ida[d] : 0;

  if (tid < 8) red[tid] = 0;
  int orpar = 0;
  __syncthreads();
  // A CU issues from its OLDEST workgroup first, so its four resident workgroups finish one after the other (11 us
  // apart at S=25, K=64) and the last one runs alone at the end.  A workgroup that is ahead steps back (s_setprio):
  // priority 3 while three or more games remain, 2 for the last but one, 1 for the first half of the last game, 0 for
  // its tiles and verdict -- the laggards then overtake, and the four finish within ~4 us.
  int games_left = static_cast<int>((a.B - 1 - static_cast<int64_t>(blockIdx.x)) / gridDim.x) + 1;
  for (int64_t g = blockIdx.x; g < a.B; g += gridDim.x, --games_left) {
    if (games_left >= 3) __builtin_amdgcn_s_setprio(3);
    else if (games_left == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(1);
    // ---- 1. factors (u negated: the products are subtracted), transposed into LDS; X0 into the image ----
    const __amdgpu_buffer_rsrc_t tok = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int8_t*>(a.actions + g * R * G::A3), 0, R * G::A3, 0x00027000);
    // (the state is requested first: tokens and state then share ONE memory round trip)
    constexpr int NCP = (G::NCHUNK + kBlock - 1) / kBlock;  // image chunks per thread
    uint4 sq[NCP];
    {
      const int8_t* src = a.in + g * a.in_stride;
      if (G::TAIL != 0 && g == a.B - 1) {  // uniform: only the batch's last game may lack the bytes behind its tail
#pragma unroll
        for (int i = 0; i < NCP; ++i) {  // (chunks past the end shadow the last one: no conditionally written registers)
          const int c = tid + kBlock * i, cc = c < G::NCHUNK ? c : G::NCHUNK - 1;
          sq[i] = load_chunk<G::TAIL>(src + 16 * cc, cc == G::NCHUNK - 1);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NCP; ++i) {
          const int c = tid + kBlock * i, cc = c < G::NCHUNK ? c : G::NCHUNK - 1;
          sq[i] = *reinterpret_cast<const uint4*>(src + 16 * cc);
        }
      }
    }
    int big = 0;
    if (srg < NRG) {
      for (int rb = 4 * srg; rb < Rp; rb += 4 * NRG * TB) {
        int f[TB][4];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int r = rb + 4 * NRG * tb + t;
            f[tb][t] = static_cast<int8_t>(__builtin_amdgcn_raw_buffer_load_b8(tok, sx + r * G::A3, 0, 0));
          }
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
          const int r0 = rb + 4 * NRG * tb;
          if (r0 < Rp) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              f[tb][t] = (r0 + t < R) ? f[tb][t] - a.shift : 0;
              big |= (f[tb][t] < slo) | (f[tb][t] > shi);
              if (sx < S) f[tb][t] = -f[tb][t];
            }
            *reinterpret_cast<uint32_t*>(T + sx * RS + r0) = pack4(f[tb][0], f[tb][1], f[tb][2], f[tb][3]);
          }
        }
      }
    }
    {
#pragma unroll
      for (int i = 0; i < NCP; ++i) {
        const int c = tid + kBlock * i;
        if (c < G::NCHUNK) *reinterpret_cast<uint4*>(img + 16 * c) = sq[i];
      }
    }
    if (tid < 4) red[tid] = 0;
    // workgroup OR of `big` with ONE barrier (HIP's __syncthreads_or costs three): red[4 + parity] was cleared during
    // the previous game, is OR-ed here, read after the barrier; the other word is cleared for the next game
    if (big) atomicOr(reinterpret_cast<unsigned*>(&red[4 + orpar]), 1u);
    __syncthreads();
    const bool anybig = red[4 + orpar] != 0;
    if (tid == 0) red[4 + (orpar ^ 1)] = 0;
    orpar ^= 1;
    if (anybig) {  // factors beyond the byte products: the lattice kernels take this game
      if (tid == 0) {
        a.done_step[g] = kNeedsExact;
        atomicAdd(&g_many_handovers, 1ull);
      }
      __syncthreads();
      continue;
    }

    // ---- 2a. per-action scalars: wavefront x < 3 takes factor vector x of every action ----
    if (wave < 3) {
      const int uw = __builtin_amdgcn_readfirstlane(wave);  // provably uniform: the weights come by scalar loads
      const int* wu0 = uw == 2 ? g_fw.w[0] : g_fw.uv[0][uw];
      const int* wu1 = uw == 2 ? g_fw.w[1] : g_fw.uv[1][uw];
      for (int r = lane; r < Rp; r += 64) {
        // (two accumulators per sum, running max and min instead of |b|: the dependent chains are half as long)
        int d0a = 0, d0b = 0, d1a = 0, d1b = 0, bmx = 0, bmn = 0;
        const int8_t* colp = T + (uw * S) * RS + r;
#pragma unroll
        for (int i = 0; i < S; ++i) {
          const int b = colp[i * RS];
          if (i & 1) {
            d0b = mad24_sgpr(wu0[i], b, d0b);
            d1b = mad24_sgpr(wu1[i], b, d1b);
          } else {
            d0a = mad24_sgpr(wu0[i], b, d0a);
            d1a = mad24_sgpr(wu1[i], b, d1a);
          }
          bmx = max(bmx, b);
          bmn = min(bmn, b);
        }
        const int mx = max(bmx, -bmn);
        sdot[(0 * 3 + wave) * Rp + r] = d0a + d0b;
        sdot[(1 * 3 + wave) * Rp + r] = d1a + d1b;
        smx[wave * Rp + r] = mx;
      }
    }
    __syncthreads();
    // every wavefront: sum_r mu_r mv_r mw_r.  Above 127 the overflow bound cannot hold whatever the final state
    // is: hand the game over now, before the expensive part
    int bound = 0;
    for (int r0 = 0; r0 < Rp; r0 += 64) {
      const int r = r0 + lane;
      int pb = 0;
      if (r < R) pb = min(smx[r] * smx[Rp + r], 1 << 12) * smx[2 * Rp + r];  // <= 2^20: the sum cannot wrap
      bound += static_cast<int>(__builtin_amdgcn_readlane(static_cast<int>(wave_inclusive_scan(static_cast<uint32_t>(pb))), 63));
      if (bound > (1 << 24)) bound = 1 << 24;
    }
    const bool wide = bound > 127;  // workgroup-uniform: the scalar bound cannot certify this game
    if (wide) {
      // ---- elementwise bound: Bnd = |X0| + sum_r |u_r| (x) |v_r| (x) |w_r|, the same tiles on absolute values ----
      for (int e = 16 * tid; e < G::TROWS * RS; e += 16 * kBlock) {
        const uint4 q = *reinterpret_cast<const uint4*>(T + e);
        *reinterpret_cast<uint4*>(Tabs + e) = uint4{abs4_i8(q.x), abs4_i8(q.y), abs4_i8(q.z), abs4_i8(q.w)};
      }
      __syncthreads();
      int bmax = 0;
      uint32_t m128 = 0;  // a byte 0x80 in |X0|: the start state holds -128, whose bound is 128 > 127 anyway
#pragma unroll 1
      for (int k = 0; k < TPW; ++k) {
        if (G::NT % NW != 0 && wave + NW * k >= G::NT) break;  // wave-uniform
        v16i acc;
#pragma unroll
        for (int t2 = 0; t2 < 16; ++t2) acc[t2] = 0;
        {
          const int m = xoff[k] & 3;
          const uint32_t* p4 = reinterpret_cast<const uint32_t*>(img + (xoff[k] & ~3));
          uint32_t d[5];
#pragma unroll
          for (int t = 0; t < 5; ++t) d[t] = p4[t];
          v4i xf;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const uint32_t ax = abs4_i8(__builtin_amdgcn_alignbyte(d[t + 1], d[t], static_cast<uint32_t>(m)));
            // only bytes k = 16 h + 4 t + b < S belong to this lane's column (the rest meets zero rows of the identity
            // and may lie beyond the image)
            const int nvb = S - (16 * h + 4 * t);
            const uint32_t vb = nvb >= 4 ? 0x80808080u : (nvb <= 0 ? 0u : (0x80808080u >> (8 * (4 - nvb))));
            if (ncol[k] >= 0) m128 |= ax & vb;
            xf[t] = static_cast<int>(ax);
          }
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(idp, xf, acc, 0, 0, 0);
        }
        for (int k0 = 0; k0 < Rp; k0 += 32) {
          const v4i w = *reinterpret_cast<const v4i*>(Tabs + woff + k0);
          const v4i p = bytemul16(*reinterpret_cast<const v4i*>(Tabs + uoff[k] + k0),
                                  *reinterpret_cast<const v4i*>(Tabs + voff[k] + k0));
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, p, acc, 0, 0, 0);
        }
        if (ncol[k] >= 0) {
#pragma unroll
          for (int t2 = 0; t2 < 16; t2 += 2) bmax = max(max(acc[t2], acc[t2 + 1]), bmax);
        }
      }
      if (m128) bmax = 128;
      bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x111, 0xf, 0xf, false));
      bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x112, 0xf, 0xf, false));
      bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x114, 0xf, 0xf, false));
      bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x118, 0xf, 0xf, false));
      bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x142, 0xa, 0xf, false));
      bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x143, 0xc, 0xf, false));
      if (lane == 63 && bmax) atomicMax(&red[3], bmax);  // (one lane per wavefront: see the reductions after the tiles)
      __syncthreads();
      if (red[3] > 127) {  // workgroup-uniform: some prefix may leave int8 -- the lattice kernels decide exactly
        if (tid == 0) {
          a.done_step[g] = kNeedsExact;
          atomicAdd(&g_many_handovers, 1ull);
        }
        __syncthreads();
        continue;
      }
    }

    if (games_left == 1) __builtin_amdgcn_s_setprio(0);
    // ---- 2b. column tiles: acc = I X0 + W P ----
    int hi = 0, lo = 0;
    uint32_t hx0 = 0, hx1 = 0;
    v4i wa[KS ? KS : 1];
#pragma unroll
    for (int k = 0; k < KS; ++k) wa[k] = *reinterpret_cast<const v4i*>(T + woff + 32 * k);
    auto tile = [&](int k, uint32_t (&X)[4]) {
      v16i acc;
#pragma unroll
      for (int t2 = 0; t2 < 16; ++t2) acc[t2] = 0;
      {  // X0 fragment: 16 bytes at image offset xoff (any alignment) from aligned dwords
        const int m = xoff[k] & 3;
        const uint32_t* p4 = reinterpret_cast<const uint32_t*>(img + (xoff[k] & ~3));
        uint32_t d[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) d[t] = p4[t];
        v4i xf;
#pragma unroll
        for (int t = 0; t < 4; ++t) xf[t] = static_cast<int>(__builtin_amdgcn_alignbyte(d[t + 1], d[t], static_cast<uint32_t>(m)));
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ida, xf, acc, 0, 0, 0);
      }
      if constexpr (KS != 0) {
        v4i p[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          p[ks] = bytemul16(*reinterpret_cast<const v4i*>(T + uoff[k] + 32 * ks),
                            *reinterpret_cast<const v4i*>(T + voff[k] + 32 * ks));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa[ks], p[ks], acc, 0, 0, 0);
      } else {
        for (int k0 = 0; k0 < Rp; k0 += 32) {
          const v4i w = *reinterpret_cast<const v4i*>(T + woff + k0);
          const v4i p = bytemul16(*reinterpret_cast<const v4i*>(T + uoff[k] + k0),
                                  *reinterpret_cast<const v4i*>(T + voff[k] + k0));
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, p, acc, 0, 0, 0);
        }
      }
      // rows S, S+1 of the lower half-wave: sum_l pw[l] X0[n][l] for this lane's column
      if (h == 0 && ncol[k] >= 0) {
        const int n = 32 * (wave + NW * k) + col;
        hx0 += static_cast<uint32_t>(mul24_pinned(static_cast<int>(cw[n]), acc[kF0]));  // |cw| < 2^22, |t| < 2^20
        hx1 += static_cast<uint32_t>(mul24_pinned(static_cast<int>(cw[G::NT * 32 + n]), acc[kF1]));
      }
      acc[kF0] = 0;
      acc[kF1] = 0;
#pragma unroll
      for (int t2 = 0; t2 < 16; t2 += 2) {
        hi = max(max(acc[t2], acc[t2 + 1]), hi);
        lo = min(min(acc[t2], acc[t2 + 1]), lo);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) X[q] = pack4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
      asm volatile("" : "+v"(hi), "+v"(lo), "+v"(hx0), "+v"(hx1), "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]));
    };
#pragma unroll
    for (int k = 0; k < TPW; k += 2) {
      if (G::NT % NW != 0 && wave + NW * k >= G::NT) break;
      __builtin_amdgcn_sched_barrier(0);
      uint32_t XA[4], XB[4] = {0, 0, 0, 0};
      tile(k, XA);
      const bool pair = k + 1 < TPW && (G::NT % NW == 0 || wave + NW * (k + 1) < G::NT);
      if (pair) tile(k + 1, XB);
      uint32_t E[9];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const auto sw = __builtin_amdgcn_permlane32_swap(XA[q], XB[q], false, false);
        E[2 * q] = sw[0];
        E[2 * q + 1] = sw[1];
      }
      E[8] = 0;
      const int nown = h ? (pair ? ncol[k + 1 < TPW ? k + 1 : k] : -1) : ncol[k];
      constexpr int Q = (S + 3) / 4, r4 = S % 4;
      static_assert(r4 <= 1, "tg_mfma.h: column emission handles S % 4 in {0, 1}");
      if constexpr (r4 == 1) {
        const uint32_t nxt = __builtin_amdgcn_update_dpp(0u, E[0], 0x130, 0xf, 0xf, false);
        E[Q - 1] |= nxt << 8;
      }
      if (nown >= 0) {  // in place over the X0 bytes this pair has consumed (tiles are 16-byte aligned blocks)
        const int base = nown * S;
        const int o = (4 - (base & 3)) & 3;
        uint8_t* dst = img + base + o;
#pragma unroll
        for (int jd = 0; jd < Q; ++jd) {
          if (4 * jd + o < S) {
            const uint32_t d = __builtin_amdgcn_alignbyte(E[jd + 1], E[jd], static_cast<uint32_t>(o));
            *reinterpret_cast<uint32_t*>(dst + 4 * jd) = d;
          }
        }
      }
    }
    // Reduce inside the wavefront first (DPP), then ONE lane per wavefront touches the LDS word: 64 lanes doing an LDS
    // atomic on the same address serialise -- the three atomics of all four wavefronts cost 5 200 cycles per game here
    // (a quarter of the game's time; fine stamps, NOTES.md), found only after every other phase had been suspected.
    {
      const uint32_t s0 = wave_inclusive_scan(hx0), s1 = wave_inclusive_scan(hx1);  // lane 63: the wavefront's sums
      int mabs = max(hi, -lo);
      mabs = max(mabs, __builtin_amdgcn_update_dpp(0, mabs, 0x111, 0xf, 0xf, false));  // row_shr:1
      mabs = max(mabs, __builtin_amdgcn_update_dpp(0, mabs, 0x112, 0xf, 0xf, false));  // row_shr:2
      mabs = max(mabs, __builtin_amdgcn_update_dpp(0, mabs, 0x114, 0xf, 0xf, false));  // row_shr:4
      mabs = max(mabs, __builtin_amdgcn_update_dpp(0, mabs, 0x118, 0xf, 0xf, false));  // row_shr:8
      mabs = max(mabs, __builtin_amdgcn_update_dpp(0, mabs, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
      mabs = max(mabs, __builtin_amdgcn_update_dpp(0, mabs, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
      if (lane == 63) {
        if (s0) atomicAdd(reinterpret_cast<unsigned*>(&red[0]), s0);
        if (s1) atomicAdd(reinterpret_cast<unsigned*>(&red[1]), s1);
        if (mabs) atomicMax(&red[2], mabs);
      }
    }
    __syncthreads();  // image, action scalars and reductions complete

    // ---- 3. verdict (every wavefront computes it: no further exchange) ----
    bool redo;
    int dstep;
    {
      const uint32_t h0 = static_cast<uint32_t>(red[0]), h1 = static_cast<uint32_t>(red[1]);
      const int maxfinal = red[2];
      uint32_t carry0 = 0, carry1 = 0;
      int first = -1;
      for (int r0 = 0; r0 < Rp; r0 += 64) {
        const int r = r0 + lane;
        uint32_t g0 = 0, g1 = 0;
        if (r < R) {
          g0 = static_cast<uint32_t>(sdot[0 * Rp + r]) * static_cast<uint32_t>(sdot[1 * Rp + r]) * static_cast<uint32_t>(sdot[2 * Rp + r]);
          g1 = static_cast<uint32_t>(sdot[3 * Rp + r]) * static_cast<uint32_t>(sdot[4 * Rp + r]) * static_cast<uint32_t>(sdot[5 * Rp + r]);
        }
        g0 = wave_inclusive_scan(g0);
        g1 = wave_inclusive_scan(g1);
        const bool cand = r < R && (h0 + carry0 + g0) == 0u && (h1 + carry1 + g1) == 0u;
        const unsigned long long mask = __ballot(cand);
        if (first < 0 && mask) first = r0 + __builtin_ctzll(mask);
        carry0 += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(g0), 63));
        carry1 += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(g1), 63));
      }
      const bool final_zero = maxfinal == 0;
      redo = !wide && maxfinal + bound > 127;                  // a step may have left int8 (wide: certified elementwise above)
      redo |= first >= 0 && first < R - 1;                     // a candidate that the final state cannot confirm
      redo |= final_zero && first != R - 1;                    // cannot happen; never trust it silently
      dstep = (first == R - 1 && final_zero) ? R - 1 : -1;
    }
    if (!redo) {
      int8_t* out = a.out + g * a.out_stride;
      for (int c = tid; c < G::NCHUNK; c += kBlock)
        store_chunk<G::TAIL>(out + 16 * c, *reinterpret_cast<const uint4*>(img + 16 * c), c == G::NCHUNK - 1);
    }
    if (tid == 0) {
      a.done_step[g] = redo ? kNeedsExact : dstep;
      if (redo) atomicAdd(&g_many_handovers, 1ull);
    }
    __syncthreads();
  }
}
