/*
 * tensor_game.h -- C ABI of libtensorgame.so: the MI355X (gfx950) tensor-game hot path.
 *
 * The reference (kurtosis/mat_mul, pure Python/PyTorch) has no FFI boundary for this path
 * (SURVEY.md section 8b): the path is reached by direct Python calls on torch tensors.  Each entry
 * point below therefore names the reference *function* it replaces (file:line in /root/reference).
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer (HIP, gfx950) unless the
 *     parameter is documented "host";
 *   - asynchronous: work is enqueued on `stream` (a hipStream_t passed as void*, NULL = the
 *     null stream); the caller synchronises.  No allocation and no host sync inside a call -- every
 *     call may be captured into a hipGraph (tg_debug_fallbacks excepted).  Thread-safe for
 *     concurrent callers: the library reads no environment variable and keeps no mutable host state
 *     besides per-device caches of device constants (CU count, workgroups per CU of a kernel), which
 *     are relaxed atomics -- two threads racing on a cold entry store the same value -- and a
 *     thread-local error string;
 *   - return 0 on success, a negative TG_ERR_* otherwise; tg_last_error() returns a thread-local
 *     message for the last failing call on this thread.  No exception crosses the boundary;
 *   - states are int8, C-contiguous (S,S,S) per game, game b at `base + b*game_stride_bytes`
 *     (game_stride_bytes >= S*S*S).  Any alignment is accepted; base and stride multiples of 16
 *     take the fast path.  Element [i][j][l] of an action tensor is u_i*v_j*w_l.
 *   - actions are int8 tokens, C-contiguous (...,3*S) = cat(u,v,w)+shift (reference utils.py:56-66);
 *     factor value = token - shift, computed in 32-bit.
 *   - the reference computes in float32 and can never overflow; here every result is computed in
 *     32-bit, narrowed to int8 with two's-complement wrap, and `overflow` (uint8 per game, may be
 *     NULL) is SET to 1 when any entry left [-128,127].  It is sticky: calls never clear it.
 *   - in-place operation (state_out == state_in) is allowed wherever both appear.
 *   - supported sizes: 1 <= S <= TG_MAX_S.
 */
#ifndef TENSOR_GAME_H_
#define TENSOR_GAME_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_ABI_VERSION 4 /* 2: the generator draws 16-bit uniforms (8 per Philox block); tg_copy_i8.  3 (additions only, round 3):
                          * tg_step_tracked_i8, tg_step_emit, tg_expand_keyed_i8, tg_seen_u64.  4 (round 4): tg_step_stream_capacity;
                          * tg_step_stream_i8 refuses ready words beyond the resident batch at S = 16 / 25 and waits 1.0 s, in time */
#define TG_MAX_S 32
#define TG_MAX_VALUES 8 /* categories of the factor distribution */

enum {
  TG_OK = 0,
  TG_ERR_INVALID = -1,     /* null pointer, size out of range, stride < S^3, misaligned actions ... */
  TG_ERR_UNSUPPORTED = -2, /* valid request this build does not implement */
  TG_ERR_HIP = -3          /* HIP runtime error (message holds hipGetErrorString) */
};

typedef void* tg_stream_t; /* hipStream_t */

int tg_abi_version(void);
const char* tg_last_error(void);
/* Debug: number of workgroups so far (this process, current device) that left the packed 16-bit
 * fast path for the exact byte-wise form.  SYNCHRONISES the device; not for hot loops. */
int tg_debug_fallbacks(uint64_t* count);
/* Debug: number of games so far that the matrix-core pass of tg_step_many_i8 could not certify (a step may have
 * left int8, or the zero state was reached before the last step) and handed to the exact lattice kernels.
 * SYNCHRONISES the device; not for hot loops. */
int tg_debug_handovers(uint64_t* count);

/* ---- the env step ------------------------------------------------------------------------- */

/* state_out[b] = state_in[b] - u(x)v(x)w of actions[b];  done[b] = (state_out[b] == 0 everywhere).
 * Replaces get_child_states (act.py:266-275) for k=1,T=1 + tensor_factorized on each game's head
 * (utils.py:181-188 as called at act.py:177).  actions: int8 (B,3S).  done: uint8 (B). */
int tg_step_i8(const int8_t* state_in, int8_t* state_out, const int8_t* actions, uint8_t* done,
               uint8_t* overflow, int64_t B, int S, int64_t game_stride_bytes, int shift,
               tg_stream_t stream);

/* The in-place step that reads only what the action touches (round 3).  tg_step_i8 must read every byte of a game to
 * answer "is it all zero?"; here the caller carries nnz[b] (int32, EXACT number of non-zero entries of game b on entry:
 * tg_done_i8 computes it, a reset knows it) and the step updates it, so only the chunks whose rows have u_i v_j != 0 are
 * loaded and stored, and done[b] = (nnz[b] == 0).  Same state, done and overflow as tg_step_i8(state, state, ...).
 * Replaces get_child_states k=1,T=1 + tensor_factorized per game (act.py:266-275, utils.py:181-188) for an env that keeps
 * its games resident.  S = 16 and S = 25 take the sparse kernels (aligned states; S = 25 from 2 048 games on, below that the
 * full step with a count); every other shape runs tg_step_i8 + the count inside this call. */
int tg_step_tracked_i8(int8_t* state, const int8_t* actions, int32_t* nnz, uint8_t* done, uint8_t* overflow,
                       int64_t B, int S, int64_t game_stride_bytes, int shift, tg_stream_t stream);

/* K sequential steps with the state resident on chip.  actions: int8 (B,K,3S).
 * done_step[b] (int32) = first step index whose post-state is all zero, or -1.
 * Replaces SyntheticDemoDataset._take_actions (datasets.py:144-153) / K calls of tg_step_i8. */
int tg_step_many_i8(const int8_t* state_in, int8_t* state_out, const int8_t* actions,
                    int32_t* done_step, uint8_t* overflow, int64_t B, int S, int K,
                    int64_t game_stride_bytes, int shift, tg_stream_t stream);

/* K in-place steps in ONE launch, for action blocks that become available step by step -- a device-side producer
 * (the policy network choosing the next action from the state, act.py:182-183 / training.py:249-255) or a host
 * thread that releases steps through copies / fills enqueued on another stream.  ready, progress, status, actions and
 * done are DEVICE memory: the stepper polls and publishes with agent-scope (sc1) accesses, which do not order against
 * a host thread writing through mapped memory.  Same results as K calls of tg_step_i8(state, state, actions[k], done[k], ...);
 * what it removes is the dependent-launch boundary between two steps (1.55 us on MI355X, more than the step itself
 * at S=4, B=65 536): the stepper stays resident, keeps every game's state on chip (in registers) and
 * per step only reads the token bytes and writes done[k] and -- at the latest when it publishes -- the new state through.
 * A wavefront takes all the steps it finds released at once (up to 8): with a producer that runs ahead the state and the
 * progress word advance in blocks of steps; a producer that releases step k+1 only after progress k+1 sees every step.
 *   actions: int8 (K,B,3S), STEP-major.  ready: uint32 (K) or NULL; step k reads its block once ready[k] != 0
 *   (the producer -- a kernel or a copy on this device -- writes the block, then ready[k]; stream order or a release at
   agent scope makes the block visible first); NULL = every
 *   block is valid at launch.  done: uint8 (K,B), done[k][b] as tg_step_i8 would report after step k.
 *   progress: uint32 (n_units) or NULL; the games are owned by n_units wavefronts, unit u = games
 *   [u*games_per_unit, (u+1)*games_per_unit) (tg_step_stream_layout); progress[u] = k+1 says: the state after step k
 *   and done[0..k] of unit u's games are visible to other agents (write-through stores, drained).  The word is
 *   non-decreasing, ends at K, and is stored before the unit waits for a ready word that is not set yet; between two
 *   stores of it the state in memory is not defined.
 *   status: uint32 (1) or NULL, set to 1 if a wavefront gave up waiting for a ready word: it waits 1.0 s (measured on the
 *   chip's 100 MHz real-time counter, whatever the clocks do), then leaves its games at the last step it published.
 * No wavefront ever waits for another one, so the launch itself cannot deadlock; a producer that waits for the WHOLE
 * batch before releasing the next step additionally needs every unit resident at once: at S = 4
 * tg_step_stream_layout chooses the games per wavefront so that this holds and refuses batches beyond what the
 * device keeps resident (262 144 games on the 256 CUs of an MI355X: four wavefronts of 64 games per SIMD; with ready == NULL tg_step_stream_i8 takes any B: units
 * of 64 games run in rounds, progress -- if given -- has (B + 63) / 64 words); S = 16 (one wavefront per game, the 4 KiB of
 * a game in registers) holds 32 games per CU = 8 192 on 256 CUs; S = 25 (one wavefront per game, the game's 15 625
 * bytes in registers) holds 16 games per CU = 4 096 on 256 CUs.  Beyond these (tg_step_stream_capacity) S = 16 / 25 run
 * their units in rounds too, with ready == NULL only: with ready words the call is refused (TG_ERR_UNSUPPORTED) -- a later
 * round would start only when an earlier one has finished all K steps.  S = 4, S = 16 and S = 25 in this build
 * (TG_ERR_UNSUPPORTED otherwise), states 16-byte aligned, actions 4-byte aligned (S = 16: 16-byte).  This is a separate entry with its own metric: the single-step figures of tg_step_i8
 * never include it. */
int tg_step_stream_i8(int8_t* state, const int8_t* actions, uint8_t* done, uint8_t* overflow,
                      const uint32_t* ready, uint32_t* progress, uint32_t* status, int64_t B, int S, int K,
                      int64_t game_stride_bytes, int shift, tg_stream_t stream);
/* n_units (wavefronts) and games_per_unit of tg_step_stream_i8 for B games (host call, no device work). */
int tg_step_stream_layout(int64_t B, int S, int64_t* n_units, int* games_per_unit);
/* The largest B tg_step_stream_i8 accepts together with ready words on the current device: every unit resident at once
 * (host call, no device work; from the occupancy of the stepper's kernels on this device). */
int tg_step_stream_capacity(int S, int64_t* games);

/* k children per parent: state_out[b*k+i] = state_in[b] - tensor(actions[b][i]).
 * done, changed, overflow: uint8 (B,k); changed[b][i] = child differs from parent (the per-game
 * form of remove_null_actions, utils.py:191-194); changed/overflow may be NULL.
 * Replaces get_child_states (act.py:266-275) with k>1, T=1. */
int tg_expand_i8(const int8_t* state_in, int8_t* state_out, const int8_t* actions, uint8_t* done,
                 uint8_t* changed, uint8_t* overflow, int64_t B, int S, int k,
                 int64_t in_stride_bytes, int64_t out_stride_bytes, int shift, tg_stream_t stream);

/* tg_expand_i8 that also returns keys_out[b*k+i] (uint64 (B,k), 8-byte aligned; NULL = tg_expand_i8) = the
 * tg_hash_u64 key of child (b,i): what extend_tree computes per child with state_to_str (act.py:188-190) before it
 * tests the tree (tg_seen_u64).  At S = 4, 16 and 25 (aligned layouts) the key is formed while the child is in registers;
 * other sizes and layouts run the key kernel over the children inside the same call. */
int tg_expand_keyed_i8(const int8_t* state_in, int8_t* state_out, const int8_t* actions, uint8_t* done,
                       uint8_t* changed, uint8_t* overflow, uint64_t* keys_out, int64_t B, int S, int k,
                       int64_t in_stride_bytes, int64_t out_stride_bytes, int shift, tg_stream_t stream);

/* done[b] = (state[b] == 0 everywhere); nnz[b] (int32, may be NULL) = number of non-zero entries.
 * Replaces tensor_factorized (utils.py:181-188) per game and the nnz bound of training.py:266. */
int tg_done_i8(const int8_t* state, uint8_t* done, int32_t* nnz, int64_t B, int S,
               int64_t game_stride_bytes, tg_stream_t stream);

/* state_out[b] = state_in[b] (the padding between games is neither read nor written).  The reference's
 * step is functional -- get_child_states returns fresh tensors (act.py:266-275) and callers keep the
 * parent (act.py:183-195) -- so an in-place env needs a snapshot/clone of a batch of states; also the
 * measured device-copy ceiling bench.py reports beside the step. */
int tg_copy_i8(const int8_t* state_in, int8_t* state_out, int64_t B, int S, int64_t in_stride_bytes,
               int64_t out_stride_bytes, tg_stream_t stream);

/* ---- reset -------------------------------------------------------------------------------- */

/* every game <- the <n,n,n> matrix-multiplication tensor, S = n*n.
 * Replaces build_matmul_tensor(dim_t,n,n,n)[0] (utils.py:143-161). */
int tg_reset_matmul_i8(int8_t* state_out, int64_t B, int n, int64_t game_stride_bytes,
                       tg_stream_t stream);

/* every game <- the S^3-byte template `start` (device pointer): the synthetic start tensor of
 * training.py:363-392 / a caller-supplied start_tensor (datasets.py:278-280). */
int tg_reset_broadcast_i8(const int8_t* start, int8_t* state_out, int64_t B, int S,
                          int64_t game_stride_bytes, tg_stream_t stream);

/* ---- synthetic-demonstration generator ---------------------------------------------------- */

/* target_out[b] = sum_r tensor(actions[b][r]); actions: int8 (B,R,3S).  The deterministic half of
 * create_synthetic_demo (utils.py:218-232; datasets.py:127-141) and uvw_to_demo (utils.py:40-53):
 * the bit-exact parity hook against reference-drawn factors. */
int tg_gen_from_factors_i8(const int8_t* actions, int8_t* target_out, uint8_t* overflow,
                           int64_t B, int S, int R, int64_t game_stride_bytes, int shift,
                           tg_stream_t stream);

/* The generator: for game id g = game_id_offset + b, for each of R terms, each of u,v,w is redrawn
 * until it is not the zero vector; tokens = cat(u,v,w)+shift -> actions_out int8 (B,R,3S);
 * target_out[b] = sum of the R rank-1 terms.  Replaces create_synthetic_demo (utils.py:203-233) /
 * SyntheticDemoDataset._create_synthetic_demos (datasets.py:124-142) + _factor_sample (:155-158).
 * RNG: Philox-4x32-10, key = seed, counter = (g_lo, g_hi, 3*r+x, attempt<<8 | block) -- keyed by
 * the GLOBAL game id, so output does not depend on how games are sharded over GPUs.  One block
 * yields EIGHT 16-bit draws: element 8*block + 2*m + half of the vector comes from output word m
 * (0..3), low half first.
 * values: host int8[n_values]; thresholds: host uint32[n_values-1], ascending cdf * 2^32:
 * a 16-bit draw d selects values[#{t : d * 2^16 >= t}] (every probability is honoured to within
 * 2^-16; tg_sample_basis_i8 draws 32-bit uniforms: d selects values[#{t : d >= t}]).
 * basis (may be NULL): int8 (B,3,S,S) per-game matrices (A,B,C); when given, every term is
 * emitted in the new basis: (u,v,w) -> (Au,Bv,Cw) (not in the reference). */
int tg_gen_demos_i8(int8_t* target_out, int8_t* actions_out, uint8_t* overflow, int64_t B, int S,
                    int R, const uint32_t* thresholds, const int8_t* values, int n_values,
                    int shift, uint64_t seed, uint64_t game_id_offset, const int8_t* basis,
                    int64_t game_stride_bytes, tg_stream_t stream);

/* ---- change of basis (from the AlphaTensor paper, NOT in the reference) ---- */

/* basis_out int8 (B,3,S,S): P = L*U, L/U unit(+-1)-diagonal lower/upper triangular with
 * off-diagonal entries drawn like factor values; det P = +-1.  lower_out/upper_out (may be NULL)
 * receive L and U.  Counter = (g_lo, g_hi, 0x80000000|mode, cell/4). */
int tg_sample_basis_i8(int8_t* basis_out, int8_t* lower_out, int8_t* upper_out, int64_t B, int S,
                       const uint32_t* thresholds, const int8_t* values, int n_values,
                       uint64_t seed, uint64_t game_id_offset, tg_stream_t stream);

/* state_out[b][a][c][d] = sum_ijk A[a][i] B[c][j] C[d][k] state_in[b][i][j][k],
 * basis int32 (B,3,S,S) (int32 so that exact inverses of unimodular matrices fit). */
int tg_change_basis_i8(const int8_t* state_in, const int32_t* basis, int8_t* state_out,
                       uint8_t* overflow, int64_t B, int S, int64_t game_stride_bytes,
                       tg_stream_t stream);

/* ---- next rows ------------------------------------------------------ */

/* N1, model-input assembly.  The env keeps a T-deep history ring per game: frame slot s of game b
 * at ring + b*game_stride_bytes + s*frame_stride_bytes (int8, S^3 bytes each); the step writes the
 * new head into slot (head_slot+1) mod T, so the reference's history shift (torch.cat of
 * act.py:271-274) costs no copy.  This entry emits the (B,T,S,S,S) tensor the model consumes
 * (model.py:101-122), newest frame first: out[b][f] = float(ring[b][(head_slot - f) mod T]).
 * out: float32 (out_dtype == 0), float16 (1) or bfloat16 (2), C-contiguous; int8 values are exact in all three.  scalars (may be NULL): float32 (B,1)
 * filled with t_step (get_scalars, utils.py:22-37). */
int tg_emit_frames(const int8_t* ring, void* out, float* scalars, int out_dtype, int64_t B, int S,
                   int T, int head_slot, float t_step, int64_t frame_stride_bytes,
                   int64_t game_stride_bytes, tg_stream_t stream);

/* N1, fused: ONE env step on the history ring and the model input of the NEW state in one call -- what extend_tree
 * does between two network evaluations (get_child_states' history shift, act.py:271-274, then the (B,T,S,S,S) tensor
 * and get_scalars for fwd_infer, act.py:178-182).  The step reads ring slot head_slot and writes the new head into slot
 * (head_slot + 1) mod T (T = 1: in place); done / overflow as tg_step_i8; out[b][0] = float(new head), out[b][f] =
 * float(ring[b][(head_slot + 1 - f) mod T]); out_dtype, scalars, strides as tg_emit_frames.  The caller's head slot
 * afterwards is (head_slot + 1) mod T.  At S = 4 and S = 16 this is one kernel while the output (< 128 MiB) stays in the
 * caches (the new head and the old head are emitted from registers); other sizes, layouts and larger outputs run
 * tg_step_i8 and tg_emit_frames inside the call. */
int tg_step_emit(int8_t* ring, const int8_t* actions, void* out, float* scalars, uint8_t* done, uint8_t* overflow,
                 int out_dtype, int64_t B, int S, int T, int head_slot, float t_step, int64_t frame_stride_bytes,
                 int64_t game_stride_bytes, int shift, tg_stream_t stream);

/* N2, transposition-table key.  hash_out[b] (uint64) = H(state[b]): with the S^3 bytes zero-padded
 * to 8-byte little-endian words w_k, H = fmix64( (sum_k fmix64(w_k + (k+1)*0x9E3779B97F4A7C15))
 * ^ (S^3 * 0xC2B2AE3D27D4EB4F) ), fmix64 = the MurmurHash3 finaliser.  Equal states give equal
 * keys; replaces the 127-character string key of state_to_str (utils.py:164-169). */
int tg_hash_u64(const int8_t* state, uint64_t* hash_out, int64_t B, int S,
                int64_t game_stride_bytes, tg_stream_t stream);

/* N2, second half: the transposition table.  extend_tree keeps only the children whose key is not yet a key of the
 * tree (`c not in new_mc_tree`, act.py:188-195) and then records the expanded state's key (act.py:209-211).  `table`
 * is a caller-owned open-addressing array of uint64, `capacity` (a power of two) entries, zero-initialised = empty
 * (hipMemset); linear probing from key & (capacity-1); a key equal to 0 is stored as 0x9E3779B97F4A7C15.  Keep the
 * load factor at or below one half.
 *   fresh[i] (uint8, may be NULL when insert != 0) = 1 iff mask[i] != 0 and keys[i] was not in the table BEFORE this
 *   call (so equal keys inside one call are all fresh, like the reference's list comprehension); mask (uint8 (n), may
 *   be NULL = all ones) is typically `changed` of tg_expand_i8, which makes fresh = "not a null action and not yet in
 *   the tree".  insert != 0: afterwards every masked key is in the table (lookup and insertion are two kernels of the
 *   same call, in that order).  status (uint32 (1), may be NULL): bit 0 is SET when a key could not be recorded because
 *   the table is full.  keys and table must be 8-byte aligned. */
int tg_seen_u64(const uint64_t* keys, uint64_t* table, int64_t capacity, uint8_t* fresh, const uint8_t* mask,
                uint32_t* status, int64_t n, int insert, tg_stream_t stream);

/* N3, terminal reward.  rank_out[b] (int32) = sum over the S slices state[b][i] of the rank of the
 * S x S integer matrix, computed exactly over GF(p) for the two primes 2^26-5 and 2^26-27 (balanced residues in
 * double precision; the maximum is taken; it equals the rational rank unless a minor is divisible by both primes).
 * Replaces get_rank (utils.py:134-140: float SVD rank, summed). */
int tg_rank_i32(const int8_t* state, int32_t* rank_out, int64_t B, int S,
                int64_t game_stride_bytes, tg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TENSOR_GAME_H_ */
