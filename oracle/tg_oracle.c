/* tg_oracle.c -- plain C restatement of the reference's tensor-game arithmetic.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): a second, independent checker next to
 * oracle/tensor_game.py, and a scalar CPU timing point for bench.py.  Never linked into, loaded
 * by, or shipped with the product (mat_mul_amd).
 *
 * Every function cites the reference file:line (in /root/reference) it follows.  States are
 * int8 (S,S,S) per game, dense (game stride S^3); tokens are int8 (..,3S) = cat(u,v,w)+shift.
 * Arithmetic is 32-bit, results are narrowed to int8 with two's-complement wrap, and the
 * per-game overflow flag is raised when a value left [-128,127] -- the build's contract
 * (include/tensor_game.h), which equals the reference's float32 result whenever the flag is clear.
 *
 * Build: make -C oracle   ->  oracle/_build/libtg_oracle.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* one step for one game: state - u(x)v(x)w, element [i][j][l] = u_i v_j w_l
 * (utils.py:56-96 action_to_uvw/uvw_to_tensor; act.py:268-270 the subtraction).
 * returns 1 when the result is all zero (utils.py:181-188 on the head of this game). */
static int step_one(const int8_t* in, int8_t* out, const int8_t* tok, int S, int shift, int sign, uint8_t* ovf) {
  int nz = 0, over = 0;
  for (int i = 0; i < S; ++i) {
    const int u = tok[i] - shift;
    for (int j = 0; j < S; ++j) {
      const int uv = u * (tok[S + j] - shift);
      for (int l = 0; l < S; ++l) {
        const int e = (i * S + j) * S + l;
        const int n = in[e] - sign * uv * (tok[2 * S + l] - shift);
        over |= (n < -128) | (n > 127);
        out[e] = (int8_t)n;
        nz |= (int8_t)n != 0;
      }
    }
  }
  if (ovf && over) *ovf = 1;
  return !nz;
}

/* get_child_states k=1,T=1 (act.py:266-275) + per-game tensor_factorized (utils.py:181-188) */
void tgo_step_i8(const int8_t* in, int8_t* out, const int8_t* actions, uint8_t* done, uint8_t* overflow,
                 int64_t B, int S, int shift) {
  const int64_t N = (int64_t)S * S * S;
  for (int64_t b = 0; b < B; ++b)
    done[b] = (uint8_t)step_one(in + b * N, out + b * N, actions + b * 3 * S, S, shift, 1, overflow ? overflow + b : 0);
}

/* SyntheticDemoDataset._take_actions (datasets.py:144-153): K sequential steps;
 * done_step = first step whose post-state is all zero, else -1 */
void tgo_step_many_i8(const int8_t* in, int8_t* out, const int8_t* actions, int32_t* done_step, uint8_t* overflow,
                      int64_t B, int S, int K, int shift) {
  const int64_t N = (int64_t)S * S * S;
  for (int64_t b = 0; b < B; ++b) {
    int8_t* cur = out + b * N;
    memmove(cur, in + b * N, (size_t)N);
    done_step[b] = -1;
    for (int k = 0; k < K; ++k) {
      const int z = step_one(cur, cur, actions + (b * K + k) * 3 * S, S, shift, 1, overflow ? overflow + b : 0);
      if (z && done_step[b] < 0) done_step[b] = k;
    }
  }
}

/* get_child_states k>1 (act.py:266-275) + per-game remove_null_actions (utils.py:191-194) */
void tgo_expand_i8(const int8_t* in, int8_t* out, const int8_t* actions, uint8_t* done, uint8_t* changed,
                   uint8_t* overflow, int64_t B, int S, int k, int shift) {
  const int64_t N = (int64_t)S * S * S;
  for (int64_t b = 0; b < B; ++b)
    for (int c = 0; c < k; ++c) {
      const int8_t* tok = actions + (b * k + c) * 3 * S;
      const int64_t child = b * k + c;
      done[child] = (uint8_t)step_one(in + b * N, out + child * N, tok, S, shift, 1, overflow ? overflow + child : 0);
      if (changed) { /* action tensor != 0  <=>  none of u, v, w is the zero vector */
        int nu = 0, nv = 0, nw = 0;
        for (int s = 0; s < S; ++s) {
          nu |= tok[s] != shift;
          nv |= tok[S + s] != shift;
          nw |= tok[2 * S + s] != shift;
        }
        changed[child] = (uint8_t)(nu && nv && nw);
      }
    }
}

/* create_synthetic_demo's accumulation (utils.py:218-232; datasets.py:127-141), uvw_to_demo
 * (utils.py:40-53): target = sum_r u_r(x)v_r(x)w_r, summed wide and narrowed once */
void tgo_gen_from_factors_i8(const int8_t* actions, int8_t* target, uint8_t* overflow, int64_t B, int S, int R,
                             int shift) {
  const int64_t N = (int64_t)S * S * S;
  int32_t* acc = (int32_t*)malloc((size_t)N * sizeof(int32_t));
  for (int64_t b = 0; b < B; ++b) {
    memset(acc, 0, (size_t)N * sizeof(int32_t));
    for (int r = 0; r < R; ++r) {
      const int8_t* tok = actions + (b * R + r) * 3 * S;
      for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
          const int uv = (tok[i] - shift) * (tok[S + j] - shift);
          if (!uv) continue;
          for (int l = 0; l < S; ++l) acc[(i * S + j) * S + l] += uv * (tok[2 * S + l] - shift);
        }
    }
    int over = 0;
    for (int64_t e = 0; e < N; ++e) {
      over |= (acc[e] < -128) | (acc[e] > 127);
      target[b * N + e] = (int8_t)acc[e];
    }
    if (overflow && over) overflow[b] = 1;
  }
  free(acc);
}

/* ---- Philox-4x32-10 (Salmon et al., SC'11) and the generator stream of include/tensor_game.h ---- */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

void tgo_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  memcpy(out, ctr, 16);
  philox4x32_10(out, key[0], key[1]);
}

/* create_synthetic_demo (utils.py:203-233) / _create_synthetic_demos (datasets.py:124-142) with the
 * build's counter-based sampler: each of u,v,w is redrawn until it is not the zero vector
 * (== the reference's joint rejection in distribution).  Writes tokens (B,R,3S) and the target. */
void tgo_gen_demos_i8(int8_t* target, int8_t* actions, uint8_t* overflow, int64_t B, int S, int R,
                      const uint32_t* thr, const int8_t* values, int nv, int shift, uint64_t seed,
                      uint64_t gid0) {
  for (int64_t b = 0; b < B; ++b) {
    const uint64_t gid = gid0 + (uint64_t)b;
    for (int r = 0; r < R; ++r)
      for (int x = 0; x < 3; ++x) {
        int8_t* dst = actions + ((b * R + r) * 3 + x) * S;
        for (uint32_t attempt = 0;; ++attempt) {
          int any = 0;
          /* one Philox block = EIGHT 16-bit draws: element 8q + 2m + half from output word m, low half first;
           * a draw d16 selects values[#{t : d16 * 2^16 >= thr_t}] (include/tensor_game.h) */
          for (int q = 0; 8 * q < S; ++q) {
            uint32_t c[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)(3 * r + x), (attempt << 8) | (uint32_t)q};
            philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
            for (int t = 0; t < 8 && 8 * q + t < S; ++t) {
              const uint64_t d = (uint64_t)((c[t >> 1] >> (16 * (t & 1))) & 0xFFFFu) << 16;
              int idx = 0;
              for (int i = 0; i < nv - 1; ++i) idx += d >= thr[i];
              const int f = values[idx];
              any |= f != 0;
              dst[8 * q + t] = (int8_t)(f + shift);
            }
          }
          if (any || attempt + 1 >= (1u << 16)) break;
        }
      }
  }
  tgo_gen_from_factors_i8(actions, target, overflow, B, S, R, shift);
}

/* tg_hash_u64 (include/tensor_game.h): the transposition key replacing state_to_str (utils.py:164-169) */
static uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xFF51AFD7ED558CCDull;
  k ^= k >> 33; k *= 0xC4CEB9FE1A85EC53ull;
  k ^= k >> 33;
  return k;
}

void tgo_hash_u64(const int8_t* state, uint64_t* out, int64_t B, int S) {
  const int64_t N = (int64_t)S * S * S, nword = (N + 7) / 8;
  for (int64_t b = 0; b < B; ++b) {
    uint64_t h = 0;
    for (int64_t k = 0; k < nword; ++k) {
      uint64_t w = 0;
      for (int t = 0; t < 8 && 8 * k + t < N; ++t) w |= (uint64_t)(uint8_t)state[b * N + 8 * k + t] << (8 * t);
      h += fmix64(w + (uint64_t)(k + 1) * 0x9E3779B97F4A7C15ull);
    }
    out[b] = fmix64(h ^ ((uint64_t)N * 0xC2B2AE3D27D4EB4Full));
  }
}

/* build_matmul_tensor(1,n,n,n)[0] (utils.py:143-161), same index formula */
void tgo_matmul_tensor_i8(int8_t* out, int n) {
  const int S = n * n;
  memset(out, 0, (size_t)S * S * S);
  for (int ik = 0; ik < n * n; ++ik)
    for (int j = 0; j < n; ++j) out[(((ik / n) * n + j) * S + (j * n + ik % n)) * S + ik] = 1;
}
