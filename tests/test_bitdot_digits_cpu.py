"""Host half of the int8 carrier sums (pyseer_amd/csrc/bitdot_i8.hip): the digit table shk_bitdot_i8_digits writes, walked in numpy the way
the kernel's lanes walk it (A-tile row -> (column, digit), K slots of a lane, C/D register -> row), recombines to the exact integer sums.
No GPU: the matrix instruction itself is replaced by an integer dot product (tests/test_bitdot_i8_gpu.py runs the real one)."""
import ctypes as C
import os

import numpy as np
import pytest

from test_bitdot_i8_gpu import _exact_sums

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyseer_amd", "libseerhip.so")


@pytest.mark.parametrize("N,NE", [(130, 9), (64, 2), (257, 12), (100, 16)])
def test_digit_table_walked_like_the_kernel_gives_the_exact_sums(N, NE):
    lib = C.CDLL(LIB)
    lib.shk_bitdot_i8_table_bytes.restype = C.c_int64
    lib.shk_bitdot_i8_table_bytes.argtypes = [C.c_int, C.c_int]
    lib.shk_bitdot_i8_digits.restype = None
    lib.shk_bitdot_i8_digits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(N + NE)
    V = 48
    NB64, NT = (N + 63) // 64, (NE + 3) // 4
    K = (rng.random((V, N)) < rng.uniform(0, 1, V)[:, None]).astype(np.uint8)
    vals = rng.standard_normal((N, NE)) * 10.0 ** rng.integers(-6, 7, NE)[None, :]
    vals[0, 0] = -np.abs(vals[:, 0]).max() * 1.75
    atab = np.zeros(lib.shk_bitdot_i8_table_bytes(NE, NB64), dtype=np.int8); scale = np.zeros(4 * NT)
    vc = np.ascontiguousarray(vals)
    lib.shk_bitdot_i8_digits(vc.ctypes.data, N, NE, NB64, atab.ctypes.data, scale.ctypes.data)
    A = atab.reshape(-1, NT, 64, 16).astype(np.int64)                 # K-step, tile, lane, byte
    assert A.shape[0] % 4 == 0 and A.shape[0] >= 2 * NB64            # whole LDS stages; steps past the last word are zero
    Kp = np.zeros((V, A.shape[0] * 32), dtype=np.int64); Kp[:, :N] = K
    acc = np.zeros((NT, 32, V), dtype=np.int64)                       # D[tile][row][variant]
    for ks in range(A.shape[0]):
        for lane in range(64):
            i, h = lane & 31, lane >> 5
            samples = ks * 32 + h * 16 + np.arange(16)                # the 16 K slots of this lane
            acc[:, i, :] += A[ks, :, lane, :] @ Kp[:, samples].T
    assert np.abs(acc).max() <= 128 * N                               # fits the int32 accumulators by a wide margin
    S, sc = _exact_sums(K, vals)
    assert np.array_equal(sc, scale[:NE])
    for t in range(NT):
        for h in range(2):
            for half in range(2):
                c = 4 * t + 2 * h + half
                for v in range(V):
                    tot = 0
                    for l in range(7, -1, -1):
                        r = half * 8 + l
                        tot = tot * 256 + int(acc[t, (r & 3) + 8 * (r >> 2) + 4 * h, v])
                    assert tot == (S[v][c] if c < NE else 0), (t, h, half, v)
