"""Thin object layer over the C ABI: one Engine per (device, n_samples).

Host arrays go through sh_*_batch (copy in / copy out); torch tensors that already live in HBM go
through sh_*_batch_dev on torch's current stream (torch is used only for device memory and streams).
"""
import ctypes as C

import numpy as np

from . import _abi
from .packing import pack_variants, row_bytes_for


def _dp(a):
    return a.ctypes.data_as(_abi.c_dp)


class Engine(object):
    def __init__(self, n_samples, device=0):
        self._lib = _abi.load()
        if self._lib.sh_device_count() <= 0:
            raise _abi.SeerHipError(_abi.SH_ENODEV, "no HIP device visible; libseerhip has no CPU fallback")
        h = self._lib.sh_create(int(device), int(n_samples))
        if not h:
            raise _abi.SeerHipError(_abi.SH_ENODEV, self._lib.sh_last_error().decode())
        self._h = C.c_void_p(h)
        self.n = int(n_samples)
        self.device = int(device)
        self.q = None
        self._keep = []
        self._inflight = (None,)           # result arrays of pipelined batches the library still writes into
        self._announced = (None,)          # rows announced with prefetch(): alive until the batch that consumes them has been called

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- common ------------------------------------------------------------------------------
    def set_stream(self, stream_handle):
        _abi.check(self._lib.sh_set_stream(self._h, C.c_void_p(int(stream_handle or 0))))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        _abi.check(self._lib.sh_synchronize(self._h))

    def set_timing(self, on=True):
        _abi.check(self._lib.sh_set_timing(self._h, int(bool(on))))

    def get_timing(self):
        """(total ms of the dominant kernel, number of launches) since set_timing(True)."""
        ms = C.c_double(); n = C.c_int64()
        _abi.check(self._lib.sh_get_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_dedup(self, on=True):
        """Test each distinct presence pattern once and fan the result out (exact; see include/seerhip.h)."""
        _abi.check(self._lib.sh_set_dedup(self._h, int(bool(on))))

    def dedup_info(self):
        n = C.c_int64()
        _abi.check(self._lib.sh_dedup_info(self._h, C.byref(n)))
        return n.value

    def set_af_filter(self, min_af, max_af):
        _abi.check(self._lib.sh_set_af_filter(self._h, float(min_af), float(max_af)))

    def _bits(self, bits):
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        if bits.ndim != 2 or bits.shape[1] * 8 < self.n:
            raise AssertionError("shape missmatch between snps and Y")       # lmm_cov.py:674
        return bits

    # ---- LMM ---------------------------------------------------------------------------------
    def lmm_setup(self, U, S, y, covar, h2, continuous=False, filter_pvalue=1.0, lrt_pvalue=1.0, n_limbs=0):
        """U (n,k), S (k,), y (n,), covar (n,D) with the intercept LAST (pyseer/lmm.py:95-99)."""
        U = np.ascontiguousarray(U, dtype=np.float64); S = np.ascontiguousarray(S, dtype=np.float64)
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        covar = np.ascontiguousarray(np.asarray(covar, dtype=np.float64).reshape(y.shape[0], -1))
        if U.shape[0] != self.n or y.shape[0] != self.n or S.shape[0] != U.shape[1]:
            raise AssertionError("shape missmatch")
        rc = self._lib.sh_lmm_setup(self._h, _dp(U), _dp(S), U.shape[1], _dp(y), _dp(covar), covar.shape[1], float(h2),
                                    int(bool(continuous)), float(filter_pvalue), float(lrt_pvalue), int(n_limbs))
        if rc == _abi.SH_EH2:
            raise KeyError("beta")                                           # lmm_test.py:416-417
        _abi.check(rc)

    def lmm_share_from(self, other):
        """Take the per-run LMM state of `other` (an Engine after lmm_setup, any device) by a device-to-device copy."""
        _abi.check(self._lib.sh_lmm_share(self._h, other._h))

    def lmm_info(self):
        nl = C.c_int(); macs = C.c_int64(); qs = C.c_double()
        _abi.check(self._lib.sh_lmm_info(self._h, C.byref(nl), C.byref(macs), C.byref(qs)))
        en = C.c_double(); ulp = C.c_double(); tol = C.c_double(); ex = C.c_int(); bt = C.c_double(); bm = C.c_double(); nr = C.c_int64()
        _abi.check(self._lib.sh_lmm_bound(self._h, C.byref(en), C.byref(ulp), C.byref(tol), C.byref(ex), C.byref(bt), C.byref(bm),
                                          C.byref(nr)))
        est = C.c_double(); sq = C.c_int()
        _abi.check(self._lib.sh_lmm_bound_estimate(self._h, C.byref(est), C.byref(sq)))
        return dict(n_limbs=nl.value, int8_macs_per_variant=macs.value, quant_scale=qs.value, quant_err_norm=en.value, ulp=ulp.value,
                    refine_tol=tol.value, extra_limbs=ex.value, bound_rel_typical=bt.value, bound_rel_max_last_batch=bm.value,
                    refined_last_batch=nr.value, quant_err_norm_power_iteration=est.value, quant_err_norm_squarings=sq.value)

    def spectral_bound(self, A, squarings=5):
        """(certified upper bound, power-iteration estimate) of the spectral norm of the symmetric matrix A, by the two device routines
        sh_lmm_setup applies to the quantisation error (include/seerhip.h sh_spectral_bound_f32)."""
        A = np.ascontiguousarray(A, dtype=np.float32)
        up = C.c_double(); pw = C.c_double()
        _abi.check(self._lib.sh_spectral_bound_f32(self._h, A.ctypes.data_as(C.POINTER(C.c_float)), int(A.shape[0]), int(squarings),
                                                   C.byref(up), C.byref(pw)))
        return up.value, pw.value

    def set_lmm_tol(self, tol):
        """Relative bound on x^T K^-1 x above which a variant is contracted again with the extra limbs (include/seerhip.h)."""
        _abi.check(self._lib.sh_set_lmm_tol(self._h, float(tol)))

    def lmm_batch(self, bits, pipelined=False):
        """bits: (V, row_bytes) uint8 host array -> dict of host arrays (raw statistics + flags).
        pipelined=True (sh_lmm_batch_async): the call returns while the batch's last chunk is still on the device; the arrays of the
        returned dict are complete once the NEXT lmm_batch call on this engine has returned, or after wait()."""
        bits = self._bits(bits)
        V = bits.shape[0]
        o = np.empty((5, V)); fl = np.empty(V, dtype=np.uint32)           # every element is written by the call
        fn = self._lib.sh_lmm_batch_async if pipelined else self._lib.sh_lmm_batch
        if V == 0 and pipelined:
            self.wait()                                                  # an empty block still completes the one before it
        elif V:
            if pipelined:
                self._inflight = (self._inflight[-1], (o, fl))           # the library writes into them after this call has returned
            _abi.check(fn(self._h, bits.ctypes.data_as(_abi.c_u8p), bits.shape[1], V, _dp(o[0]), _dp(o[1]),
                          _dp(o[2]), _dp(o[3]), _dp(o[4]), fl.ctypes.data_as(_abi.c_u32p)))
            if not pipelined:
                self._inflight = (None,)                                 # (a synchronous call completes the batch before it too)
        return dict(prep=o[0], pvalue=o[1], beta=o[2], bse=o[3], frac_h2=o[4], flags=fl)

    def prefetch(self, bits):
        """Announce the rows of the batch after the next one (sh_prefetch_rows): call before batch k with block k+1's rows; the library uploads
        their first chunk while batch k runs.  The SAME array object must then be passed to the batch call (it is kept alive here)."""
        bits = self._bits(bits)
        self._announced = (self._announced[-1], bits)       # the one before stays alive until its batch has certainly been called
        if bits.shape[0]:
            _abi.check(self._lib.sh_prefetch_rows(self._h, bits.ctypes.data_as(_abi.c_u8p), bits.shape[1], bits.shape[0]))

    def wait(self):
        """Complete the results of the last pipelined batch (sh_wait)."""
        _abi.check(self._lib.sh_wait(self._h))
        self._inflight = (None,)

    def lmm_batch_dev(self, bits_t, out_t=None, flags_t=None):
        """bits_t: torch uint8 CUDA tensor (V, row_bytes); returns (out (5,V) float64, flags (V,) int32) CUDA tensors."""
        import torch
        V, rb = bits_t.shape
        if out_t is None:
            out_t = torch.empty((5, V), dtype=torch.float64, device=bits_t.device)
        if flags_t is None:
            flags_t = torch.empty((V,), dtype=torch.int32, device=bits_t.device)
        _abi.check(self._lib.sh_lmm_batch_dev(self._h, C.c_void_p(bits_t.data_ptr()), rb, V, C.c_void_p(out_t.data_ptr()),
                                              C.c_void_p(flags_t.data_ptr())))
        return out_t, flags_t

    # ---- fixed effects -----------------------------------------------------------------------
    def glm_setup(self, y, W, continuous, null_llf, null_firth, pret=1.0, lrtt=1.0, force_firth=False):
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        if W is None or np.size(W) == 0:
            W = np.zeros((self.n, 0))
        W = np.ascontiguousarray(np.asarray(W, dtype=np.float64).reshape(self.n, -1))
        self.q = W.shape[1]
        nf = float("nan") if null_firth is None else float(null_firth)
        Wp = W if self.q else np.zeros(1)
        _abi.check(self._lib.sh_glm_setup(self._h, _dp(y), _dp(Wp), self.q, int(bool(continuous)), float(null_llf), nf,
                                          float(pret), float(lrtt), int(bool(force_firth))))

    def glm_batch(self, bits, pipelined=False):
        """pipelined=True: as lmm_batch (sh_glm_batch_async)."""
        bits = self._bits(bits)
        V = bits.shape[0]; q = self.q
        o = np.empty((5, V)); betas = np.empty((V, max(q, 1))); fl = np.empty(V, dtype=np.uint32)   # all written by the call
        if q == 0:
            betas[:] = np.nan
        fn = self._lib.sh_glm_batch_async if pipelined else self._lib.sh_glm_batch
        if V == 0 and pipelined:
            self.wait()
        elif V:
            if pipelined:
                self._inflight = (self._inflight[-1], (o, betas, fl))
            _abi.check(fn(self._h, bits.ctypes.data_as(_abi.c_u8p), bits.shape[1], V, _dp(o[0]), _dp(o[1]),
                          _dp(o[2]), _dp(o[3]), _dp(o[4]), _dp(betas), fl.ctypes.data_as(_abi.c_u32p)))
            if not pipelined:
                self._inflight = (None,)
        return dict(prep=o[0], pvalue=o[1], kbeta=o[2], bse=o[3], intercept=o[4], betas=betas[:, :q], flags=fl)

    # ---- lineage effect ------------------------------------------------------------------------
    def lineage_setup(self, lin, cov=None):
        """lin (n, l): MDS components or cluster indicators; cov (n, j) or None (model.py:151-199)."""
        lin = np.asarray(lin, dtype=np.float64)
        if lin.ndim == 1:
            lin = lin.reshape(-1, 1)
        if lin.ndim != 2 or lin.shape[0] != self.n:                # never reshape a design of another sample count into this one
            raise AssertionError("lineage design has %s rows, the engine %d samples" % (lin.shape[0] if lin.ndim else 0, self.n))
        lin = np.ascontiguousarray(lin)
        j = 0
        covp = None
        if cov is not None and np.size(cov) and np.asarray(cov).shape[0] == self.n:
            cov = np.ascontiguousarray(np.asarray(cov, dtype=np.float64).reshape(self.n, -1)); j = cov.shape[1]; covp = _dp(cov)
        _abi.check(self._lib.sh_lineage_setup(self._h, _dp(lin), lin.shape[1], covp, j))

    def lineage_batch(self, bits):
        """-> int array of max-lineage indices, -1 where the reference returns None."""
        bits = self._bits(bits)
        out = np.full(bits.shape[0], -1, dtype=np.int32)
        if bits.shape[0]:
            _abi.check(self._lib.sh_lineage_batch(self._h, bits.ctypes.data_as(_abi.c_u8p), bits.shape[1], bits.shape[0],
                                                  out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    # ---- similarity (kinship) accumulation: pyseer/similarity.py:99-113
    def sim_begin(self):
        _abi.check(self._lib.sh_sim_begin(self._h))

    def sim_accumulate(self, bits):
        bits = self._bits(bits)
        if bits.shape[0]:
            _abi.check(self._lib.sh_sim_accumulate(self._h, bits.ctypes.data_as(_abi.c_u8p), bits.shape[1], bits.shape[0]))

    def sim_accumulate_dev(self, bits_t):
        V, rb = bits_t.shape
        _abi.check(self._lib.sh_sim_accumulate_dev(self._h, C.c_void_p(bits_t.data_ptr()), rb, V))

    def sim_finish(self):
        K = np.empty((self.n, self.n), dtype=np.float64)
        _abi.check(self._lib.sh_sim_finish(self._h, K.ctypes.data_as(_abi.c_dp)))
        return K

    def glm_info(self):
        a = C.c_int64(); b = C.c_int64()
        _abi.check(self._lib.sh_glm_info(self._h, C.byref(a), C.byref(b)))
        return dict(firth_routed=a.value, pinv_routed=b.value)

    def glm_batch_dev(self, bits_t, out_t=None, flags_t=None):
        import torch
        V, rb = bits_t.shape
        if out_t is None:
            out_t = torch.empty((5 + self.q, V), dtype=torch.float64, device=bits_t.device)
        if flags_t is None:
            flags_t = torch.empty((V,), dtype=torch.int32, device=bits_t.device)
        _abi.check(self._lib.sh_glm_batch_dev(self._h, C.c_void_p(bits_t.data_ptr()), rb, V, C.c_void_p(out_t.data_ptr()),
                                              C.c_void_p(flags_t.data_ptr())))
        return out_t, flags_t

    def glm_batch_dev_async(self, bits_t, out_t, flags_t):
        """The same batch on one of the context's lanes (sh_glm_batch_dev_async): returns at once; out_t / flags_t are complete after wait().
        The caller keeps the three tensors alive (and the rows unchanged) until then."""
        V, rb = bits_t.shape
        _abi.check(self._lib.sh_glm_batch_dev_async(self._h, C.c_void_p(bits_t.data_ptr()), rb, V, C.c_void_p(out_t.data_ptr()),
                                                    C.c_void_p(flags_t.data_ptr())))

    def set_lanes(self, n):
        _abi.check(self._lib.sh_set_lanes(self._h, int(n)))

    def get_lanes(self):
        return int(self._lib.sh_get_lanes(self._h))


class Job(object):
    """The job stream of one context (include/seerhip.h sh_job_*): submit blocks of packed rows, collect the TSV text of the rows the run
    prints and the block's (pre-filtered, tested, printed) counts.  Up to self.depth blocks may be in flight; collect() returns them in order.
    Everything a submitted block points at (rows, counts, names) is kept alive here until it has been collected."""

    def __init__(self, engine, lmm, print_filtered=False, lineage_labels=None, lineage_per_variant=False, patterns=False, sample_names=None):
        """lineage_labels: the lineage column of the output (sh_job_set_lineage; the engine's lineage_setup must have run): fixed effects fit
        every printed row, the LMM one variant per block -- its last -- as pyseer/lmm.py:209-213 does, unless lineage_per_variant.
        patterns: hash_pattern of every tested variant, on the device (patterns() after each collect).  sample_names: the run's samples in the
        engine's order -- printed rows carry their two sample lists (--print-samples), names sorted as the reference sorts them."""
        self._lib = engine._lib
        self._eng = engine
        h = self._lib.sh_job_open(engine._h, int(bool(lmm)), int(bool(print_filtered)))
        if not h:
            raise _abi.SeerHipError(_abi.SH_EINVAL, self._lib.sh_last_error().decode())
        self._h = C.c_void_p(h)
        if lineage_labels is not None:
            enc = [str(x).encode() for x in lineage_labels]
            arr = (C.c_char_p * max(len(enc), 1))(*enc)
            rc = self._lib.sh_job_set_lineage(self._h, arr, len(enc), int(bool(lineage_per_variant)))
            if rc:
                self._lib.sh_job_close(self._h); self._h = None
                _abi.check(rc)
        self._pat = bool(patterns)
        if patterns:
            _abi.check(self._lib.sh_job_set_patterns(self._h, 1))
        if sample_names is not None:
            enc = [str(x).encode() for x in sample_names]
            off = np.zeros(len(enc) + 1, dtype=np.int64)
            np.cumsum([len(x) for x in enc], out=off[1:])
            order = np.array(sorted(range(len(enc)), key=lambda i: str(sample_names[i])), dtype=np.int32)
            _abi.check(self._lib.sh_job_set_samples(self._h, b"".join(enc), off.ctypes.data, order.ctypes.data, len(enc)))
        self._ptext = C.c_void_p(); self._pn = C.c_int64()
        self.depth = int(self._lib.sh_job_depth(self._h))      # blocks in flight at most: 3 (LMM), 2 + lanes (fixed effects)
        self._held = []                       # per block in flight: the objects its pointers refer to
        self._text = C.c_void_p(); self._n = C.c_int64(); self._cnt = (C.c_int64 * 4)()

    def pending(self):
        return len(self._held)

    def submit(self, bits, counts, names_blob, name_off, rows_are_dma=False, keep=None):
        """bits (V, row_bytes) uint8, counts (V,) int32, names_blob bytes-like, name_off (V + 1,) int64: C-contiguous arrays / buffers."""
        V, rb = bits.shape
        assert bits.dtype == np.uint8 and bits.flags.c_contiguous and counts.dtype == np.int32 and counts.flags.c_contiguous
        assert name_off.dtype == np.int64 and name_off.flags.c_contiguous and counts.shape[0] == V and name_off.shape[0] == V + 1
        nb = np.frombuffer(names_blob, dtype=np.uint8) if not isinstance(names_blob, np.ndarray) else names_blob
        self._held.append((bits, counts, nb, name_off, keep))
        rc = self._lib.sh_job_submit(self._h, bits.ctypes.data, rb, V, counts.ctypes.data, nb.ctypes.data, name_off.ctypes.data, int(bool(rows_are_dma)))
        if rc:
            # (the failure may come from the PREVIOUS block's compute, after this block's upload was queued: the library's own count of blocks in
            # flight decides whether this block's buffers are still referred to; anything dropped from the list here stays alive until close())
            self._resync()
            _abi.check(rc)

    def collect(self):
        """-> (memoryview of the oldest block's text, valid until this thread's next collect; (pre-filtered, tested, printed), its keep object)"""
        rc = self._lib.sh_job_collect(self._h, C.byref(self._text), C.byref(self._n), self._cnt)
        if rc:
            self._resync()
            _abi.check(rc)
        held = self._held.pop(0)
        n = self._n.value
        text = (C.c_char * n).from_address(self._text.value) if n else b""
        return memoryview(text), (self._cnt[0], self._cnt[1], self._cnt[2]), held[4]

    def _resync(self):
        """After an error: pending() follows sh_job_pending (the C side's submitted - collected); blocks it no longer counts move to a list
        that is only dropped at close(), behind the drain of the copy stream (an upload in flight may still read them)."""
        n = int(self._lib.sh_job_pending(self._h))
        if 0 <= n < len(self._held):
            drop = len(self._held) - n
            self._zombies = getattr(self, "_zombies", []) + self._held[:drop]
            self._held = self._held[drop:]

    def run_packed(self, path, part=(0, 1), block_rows=1 << 16, use_dma=True, out_fd=1, pat_fd=-1, stop=None):
        """The whole stream of part i of n of a packed cache inside the library (sh_job_run_packed): text to out_fd, patterns to pat_fd.
        stop: a ctypes.c_int another thread may set to end the stream.  -> (pre-filtered, tested, printed, blocks)."""
        cnt = (C.c_int64 * 4)()
        _abi.check(self._lib.sh_job_run_packed(self._h, str(path).encode(), int(part[0]), int(part[1]), int(block_rows), int(bool(use_dma)), int(out_fd),
                                               int(pat_fd), cnt, C.addressof(stop) if stop is not None else None))
        return int(cnt[0]), int(cnt[1]), int(cnt[2]), int(cnt[3])

    def patterns(self):
        """The pattern text (25 bytes per tested variant) of the block last collected; valid until the next collect."""
        _abi.check(self._lib.sh_job_patterns(self._h, C.byref(self._ptext), C.byref(self._pn)))
        n = self._pn.value
        return memoryview((C.c_char * n).from_address(self._ptext.value)) if n else b""

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sh_job_close(self._h)
            self._h = None
            self._held = []
            self._zombies = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


__all__ = ["Engine", "Job", "pack_variants", "row_bytes_for"]
