"""Presence vectors <-> the engine's packed bit rows (LSB-first, variant-major; include/seerhip.h)."""
import numpy as np


def row_bytes_for(n_samples):
    """Bytes per packed row, rounded up to 8 so rows stay 8-byte aligned for the repack kernel."""
    return ((n_samples + 63) // 64) * 8


def pack_variants(K):
    """K: (V, N) array of 0/1 (any dtype) -> (V, row_bytes) uint8, bit i of a row = sample i."""
    K = np.ascontiguousarray(K)
    V, N = K.shape
    rb = row_bytes_for(N)
    out = np.zeros((V, rb), dtype=np.uint8)
    packed = np.packbits(K.astype(bool), axis=1, bitorder="little")
    out[:, :packed.shape[1]] = packed
    return out


def unpack_variants(bits, n_samples):
    """Inverse of pack_variants -> (V, N) uint8."""
    return np.unpackbits(np.ascontiguousarray(bits), axis=1, bitorder="little")[:, :n_samples]
