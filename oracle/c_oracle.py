"""ctypes binding of the plain-C oracle (oracle/tg_oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "_build" / "libtg_oracle.so"


def build() -> Path:
    subprocess.run(["make", "-s", "-C", str(HERE)], check=True)
    return LIB


def load() -> C.CDLL:
    if not LIB.exists() or LIB.stat().st_mtime < (HERE / "tg_oracle.c").stat().st_mtime:
        build()
    return C.CDLL(str(LIB))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class COracle:
    def __init__(self):
        self.lib = load()

    def step_i8(self, state, tokens, shift=1):
        state = np.ascontiguousarray(state, np.int8)
        tokens = np.ascontiguousarray(tokens, np.int8)
        B, S = state.shape[0], state.shape[1]
        out, done, ovf = np.empty_like(state), np.zeros(B, np.uint8), np.zeros(B, np.uint8)
        self.lib.tgo_step_i8(_p(state), _p(out), _p(tokens), _p(done), _p(ovf), C.c_int64(B), S, shift)
        return out, done, ovf

    def step_many_i8(self, state, tokens, shift=1):
        state = np.ascontiguousarray(state, np.int8)
        tokens = np.ascontiguousarray(tokens, np.int8)
        B, S, K = state.shape[0], state.shape[1], tokens.shape[1]
        out, ds, ovf = np.empty_like(state), np.zeros(B, np.int32), np.zeros(B, np.uint8)
        self.lib.tgo_step_many_i8(_p(state), _p(out), _p(tokens), _p(ds), _p(ovf), C.c_int64(B), S, K, shift)
        return out, ds, ovf

    def expand_i8(self, state, tokens, shift=1):
        state = np.ascontiguousarray(state, np.int8)
        tokens = np.ascontiguousarray(tokens, np.int8)
        B, S, k = state.shape[0], state.shape[1], tokens.shape[1]
        out = np.empty((B, k, S, S, S), np.int8)
        done, chg, ovf = (np.zeros((B, k), np.uint8) for _ in range(3))
        self.lib.tgo_expand_i8(_p(state), _p(out), _p(tokens), _p(done), _p(chg), _p(ovf), C.c_int64(B), S, k, shift)
        return out, done, chg, ovf

    def gen_from_factors_i8(self, tokens, shift=1):
        tokens = np.ascontiguousarray(tokens, np.int8)
        B, R, S = tokens.shape[0], tokens.shape[1], tokens.shape[2] // 3
        out, ovf = np.empty((B, S, S, S), np.int8), np.zeros(B, np.uint8)
        self.lib.tgo_gen_from_factors_i8(_p(tokens), _p(out), _p(ovf), C.c_int64(B), S, R, shift)
        return out, ovf

    def gen_demos_i8(self, B, S, R, thresholds, values, shift, seed, game_id_offset=0):
        thr = np.ascontiguousarray(thresholds, np.uint32)
        val = np.ascontiguousarray(values, np.int8)
        tok, tgt, ovf = np.empty((B, R, 3 * S), np.int8), np.empty((B, S, S, S), np.int8), np.zeros(B, np.uint8)
        self.lib.tgo_gen_demos_i8(_p(tgt), _p(tok), _p(ovf), C.c_int64(B), S, R, _p(thr), _p(val), len(val), shift,
                                  C.c_uint64(seed), C.c_uint64(game_id_offset))
        return tok, tgt, ovf

    def state_hash(self, state):
        state = np.ascontiguousarray(state, np.int8)
        out = np.empty(state.shape[0], np.uint64)
        self.lib.tgo_hash_u64(_p(state), _p(out), C.c_int64(state.shape[0]), state.shape[1])
        return out

    def matmul_tensor(self, n):
        out = np.empty((n * n,) * 3, np.int8)
        self.lib.tgo_matmul_tensor_i8(_p(out), n)
        return out

    def philox(self, ctr, key):
        c, k, o = np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), np.empty(4, np.uint32)
        self.lib.tgo_philox(_p(c), _p(k), _p(o))
        return o
