"""Array-backed result sink: whole blocks of results -> TSV text through libseerhip's native formatter (csrc/writer.cpp),
replacing one format_output call per variant (pyseer/utils.py:39-105; print loop pyseer/__main__.py:805-827)."""
import ctypes as C

import numpy as np

from . import _abi


def names_blob(names):
    """list of str -> (bytes, int64 offsets) in the layout sh_reader_next / sh_format_rows use."""
    enc = [x.encode() for x in names]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        np.cumsum([len(x) for x in enc], out=off[1:])
    return b''.join(enc), off


class RowFormatter(object):
    def __init__(self, lineage_labels=None):
        self._lib = _abi.load()
        self._labels = None
        self._nlab = 0
        if lineage_labels is not None:
            enc = [str(x).encode() for x in lineage_labels]
            self._labels = (C.c_char_p * max(len(enc), 1))(*enc)
            self._nlab = len(enc)
        self._buf = C.create_string_buffer(1 << 20)

    def format(self, blob, off, sel, cols, flags, betas=None, betas_valid=None, lineage=None):
        """cols: list of float64 arrays (one value per variant of the block); sel: int64 row order to print -> bytes."""
        return bytes(self.format_view(blob, off, sel, cols, flags, betas, betas_valid, lineage))

    def format_view(self, blob, off, sel, cols, flags, betas=None, betas_valid=None, lineage=None):
        """As format(), but returns a view of the formatter's own buffer, valid until the next call: the driver writes it out at once, so the
        text is never copied into a Python bytes object (23 MB per 262 144-row block)."""
        sel = np.ascontiguousarray(sel, dtype=np.int64)
        if sel.shape[0] == 0:
            return b''
        cols = [np.ascontiguousarray(c, dtype=np.float64) for c in cols]
        flags = np.ascontiguousarray(flags, dtype=np.uint32)
        off = np.ascontiguousarray(off, dtype=np.int64)
        cp = (_abi.c_dp * len(cols))(*[c.ctypes.data_as(_abi.c_dp) for c in cols])
        q = 0
        bp = None
        vp = None
        if betas is not None and betas.shape[1] > 0:
            betas = np.ascontiguousarray(betas, dtype=np.float64)
            betas_valid = np.ascontiguousarray(betas_valid, dtype=np.uint8)
            q = betas.shape[1]
            bp = betas.ctypes.data_as(_abi.c_dp)
            vp = betas_valid.ctypes.data_as(_abi.c_u8p)
        lp = None
        if lineage is not None:
            lineage = np.ascontiguousarray(lineage, dtype=np.int32)
            lp = lineage.ctypes.data_as(C.POINTER(C.c_int32))
        while True:
            n = self._lib.sh_format_rows(blob, off.ctypes.data_as(C.POINTER(C.c_int64)), sel.ctypes.data_as(C.POINTER(C.c_int64)),
                                         sel.shape[0], cp, len(cols), bp, q, vp, lp, self._labels, self._nlab,
                                         flags.ctypes.data_as(_abi.c_u32p), self._buf, len(self._buf))
            if n >= 0:
                return memoryview(self._buf)[:n]
            if n == -1:
                raise ValueError("sh_format_rows: bad arguments")
            self._buf = C.create_string_buffer(int(-n) + (1 << 16))
