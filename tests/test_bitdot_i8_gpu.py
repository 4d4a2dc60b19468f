"""Carrier sums on the int8 matrix cores (pyseer_amd/csrc/bitdot_i8.hip) against exact integer arithmetic.

The kernel is internal (no entry of include/seerhip.h: sh_glm_batch launches it); it is called here through its `shk_` symbol with torch
holding the device buffers.  What it feeds: the bordered first Newton step and the exact part of the information matrix of the logistic
fits (pyseer/model.py:316-344 restated in csrc/glm_kernels.hip), the k-row of I in csrc/firth_fast.hip."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _exact_sums(K, vals):
    """sum over the carriers of the 62-bit fixed-point values the kernel works with, as Python integers; and the column scales"""
    N, NE = vals.shape
    m = np.abs(vals).max(axis=0)
    ex = np.where(m > 0, np.floor(np.log2(np.where(m > 0, m, 1.0))).astype(int) + 1, 0)
    for c in range(NE):                                              # ilogb(m) + 1: m < 2^ex (log2 can be off by one at powers of two)
        while m[c] >= 2.0 ** ex[c]: ex[c] += 1
        while m[c] > 0 and m[c] < 2.0 ** (ex[c] - 1): ex[c] -= 1
    X = np.empty((N, NE), dtype=object)
    for c in range(NE):
        X[:, c] = [int(np.rint(np.ldexp(x, 62 - int(ex[c])))) for x in vals[:, c]]     # rint = llrint: ties to even; exact (|X| <= 2^62 < 2^63)
    hi = np.array([[x >> 31 for x in row] for row in X], dtype=np.int64)
    lo = np.array([[x & ((1 << 31) - 1) for x in row] for row in X], dtype=np.int64)
    Ki = K.astype(np.int64)
    SH, SL = Ki @ hi, Ki @ lo                                         # exact in int64 (N * 2^32 < 2^63)
    S = [[(int(SH[v, c]) << 31) + int(SL[v, c]) for c in range(NE)] for v in range(K.shape[0])]
    return S, np.ldexp(1.0, ex - 62)


@pytest.mark.parametrize("N,NE", [(777, 12), (64, 2), (5000, 12), (1000, 5), (130, 9), (250, 16), (333, 13)])
def test_carrier_sums_equal_the_exactly_rounded_integer_sums(N, NE):
    import torch
    from pyseer_amd import _abi
    _abi.load()
    lib = C.CDLL(_abi.LIB_PATH)
    lib.shk_bitdot_i8_table_bytes.restype = C.c_int64
    lib.shk_bitdot_i8_table_bytes.argtypes = [C.c_int, C.c_int]
    lib.shk_bitdot_i8_digits.restype = None
    lib.shk_bitdot_i8_digits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.shk_bitdot_i8.restype = C.c_int
    lib.shk_bitdot_i8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(1000 * NE + N)
    V = 700 if N < 5000 else 300
    Vpad = (V + 255) // 256 * 256
    NB64 = (N + 63) // 64
    af = rng.uniform(0.0, 1.0, V); af[:3] = (0.0, 1.0, 0.5)
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    vals = rng.standard_normal((N, NE)) * 10.0 ** rng.integers(-6, 7, NE)[None, :]
    vals[:, 0] = np.abs(vals[:, 0])                                   # a weight column; the others signed
    if NE > 3:
        vals[:, 2] = 0.0                                              # an all-zero column
        vals[rng.integers(0, N, 5), 3] = 0.0
    vals[0, NE - 1] = -np.abs(vals[:, NE - 1]).max() * 1.9999999       # the largest magnitude negative, just under a power of two's double
    # bits as the engine's transposed words: T[wd * Vpad + v], bit b of word wd = sample 64 wd + b
    Kp = np.zeros((Vpad, NB64 * 64), dtype=np.uint8); Kp[:V, :N] = K
    T = np.packbits(Kp.reshape(Vpad, NB64, 64), axis=2, bitorder="little").view(np.uint64).reshape(Vpad, NB64).T.copy()
    nbytes = lib.shk_bitdot_i8_table_bytes(NE, NB64)
    atab = np.zeros(nbytes, dtype=np.int8); scale = np.zeros(4 * ((NE + 3) // 4))
    vc = np.ascontiguousarray(vals)
    lib.shk_bitdot_i8_digits(vc.ctypes.data, N, NE, NB64, atab.ctypes.data, scale.ctypes.data)
    dT = torch.from_numpy(T.view(np.int64)).cuda(); dA = torch.from_numpy(atab).cuda(); dS = torch.from_numpy(scale).cuda()
    out = torch.full((NE, Vpad), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    rc = lib.shk_bitdot_i8(None, dT.data_ptr(), Vpad, NB64, NE, dA.data_ptr(), dS.data_ptr(), out.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()[:, :V].T                                  # V x NE
    S, sc = _exact_sums(K, vals)
    assert np.array_equal(sc, scale[:NE])
    want = np.array([[float(S[v][c]) * sc[c] for c in range(NE)] for v in range(V)])      # float(int) rounds correctly; the scale is a power of two
    # Horner in fp64: exact until the value passes 2^53, one rounding per step after that (<= 3 steps): a few ulp
    err = np.abs(got - want)
    assert (err <= 4 * np.spacing(np.abs(want))).all(), (float(err.max()), np.unravel_index(err.argmax(), err.shape))
    assert np.isfinite(got).all() and (out.cpu().numpy()[:, V:] == 0).all()           # padded rows: zero bits, zero sums
    # and against the plain fp64 dot product, relative to the sum of magnitudes
    ref = K.astype(np.float64) @ vals
    mag = K.astype(np.float64) @ np.abs(vals)
    assert (np.abs(got - ref) <= 1e-13 * mag + 1e-300).all()
