"""ctypes binding of libseerhip.so (include/seerhip.h).  This is the stub a pyseer maintainer would add.

Fails loudly when the library or a gfx950 device is missing: there is no CPU fallback in the product path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEERHIP_LIB") or os.path.join(_HERE, "libseerhip.so")      # SEERHIP_LIB: A/B builds (development)

SH_OK, SH_EINVAL, SH_ENODEV, SH_ENOMEM, SH_EH2, SH_ESHAPE, SH_EHIP = 0, -1, -2, -3, -4, -5, -6

TORCH_FIRST = True

c_dp = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)
c_u32p = C.POINTER(C.c_uint32)

# every symbol include/seerhip.h declares: (restype, argtypes)
SIGNATURES = {
    "sh_abi_version": (C.c_int, []),
    "sh_last_error": (C.c_char_p, []),
    "sh_device_count": (C.c_int, []),
    "sh_warmup": (C.c_int, [C.c_int]),
    "sh_set_wait_mode": (None, [C.c_int]),
    "sh_create": (C.c_void_p, [C.c_int, C.c_int]),
    "sh_destroy": (None, [C.c_void_p]),
    "sh_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sh_synchronize": (C.c_int, [C.c_void_p]),
    "sh_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "sh_get_timing": (C.c_int, [C.c_void_p, c_dp, C.POINTER(C.c_int64)]),
    "sh_set_dedup": (C.c_int, [C.c_void_p, C.c_int]),
    "sh_dedup_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "sh_set_af_filter": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "sh_lmm_setup": (C.c_int, [C.c_void_p, c_dp, c_dp, C.c_int, c_dp, c_dp, C.c_int, C.c_double, C.c_int,
                               C.c_double, C.c_double, C.c_int]),
    "sh_lmm_batch": (C.c_int, [C.c_void_p, c_u8p, C.c_int64, C.c_int64, c_dp, c_dp, c_dp, c_dp, c_dp, c_u32p]),
    "sh_lmm_batch_async": (C.c_int, [C.c_void_p, c_u8p, C.c_int64, C.c_int64, c_dp, c_dp, c_dp, c_dp, c_dp, c_u32p]),
    "sh_wait": (C.c_int, [C.c_void_p]),
    "sh_prefetch_rows": (C.c_int, [C.c_void_p, c_u8p, C.c_int64, C.c_int64]),
    "sh_lmm_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "sh_lmm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64), c_dp]),
    "sh_lmm_share": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sh_set_lmm_tol": (C.c_int, [C.c_void_p, C.c_double]),
    "sh_lmm_bound": (C.c_int, [C.c_void_p, c_dp, c_dp, c_dp, C.POINTER(C.c_int), c_dp, c_dp, C.POINTER(C.c_int64)]),
    "sh_lmm_bound_estimate": (C.c_int, [C.c_void_p, c_dp, C.POINTER(C.c_int)]),
    "sh_spectral_bound_f32": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, c_dp, c_dp]),
    "sh_glm_setup": (C.c_int, [C.c_void_p, c_dp, c_dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                               C.c_double, C.c_int]),
    "sh_glm_batch": (C.c_int, [C.c_void_p, c_u8p, C.c_int64, C.c_int64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_u32p]),
    "sh_glm_batch_async": (C.c_int, [C.c_void_p, c_u8p, C.c_int64, C.c_int64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_u32p]),
    "sh_lineage_setup": (C.c_int, [C.c_void_p, c_dp, C.c_int, c_dp, C.c_int]),
    "sh_lineage_batch": (C.c_int, [C.c_void_p, c_u8p, C.c_int64, C.c_int64, C.POINTER(C.c_int32)]),
    "sh_sim_begin": (C.c_int, [C.c_void_p]),
    "sh_sim_accumulate": (C.c_int, [C.c_void_p, c_u8p, C.c_int64, C.c_int64]),
    "sh_sim_accumulate_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "sh_sim_finish": (C.c_int, [C.c_void_p, c_dp]),
    "sh_reader_open": (C.c_void_p, [C.c_char_p, C.POINTER(C.c_char_p), C.c_int]),
    "sh_reader_set_concurrency": (None, [C.c_int]),
    "sh_reader_close": (None, [C.c_void_p]),
    "sh_reader_error": (C.c_char_p, []),
    "sh_reader_next": (C.c_int64, [C.c_void_p, C.c_int64, c_u8p, C.c_int64, C.POINTER(C.c_int32), C.c_char_p, C.c_int64,
                                   C.POINTER(C.c_int64)]),
    "sh_reader_names_needed": (C.c_int64, [C.c_void_p]),
    "sh_reader_buffered": (C.c_int64, [C.c_void_p]),
    "sh_reader_par_chunks": (C.c_int64, [C.c_void_p]),
    "sh_format_rows": (C.c_int64, [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int64, C.POINTER(c_dp), C.c_int, c_dp,
                                   C.c_int, c_u8p, C.POINTER(C.c_int32), C.POINTER(C.c_char_p), C.c_int, c_u32p, C.c_char_p,
                                   C.c_int64]),
    "sh_glm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sh_glm_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    # the job stream (round 5): blocks in, the text of the printed rows + counters out
    "sh_job_open": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "sh_job_close": (None, [C.c_void_p]),
    "sh_job_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "sh_job_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sh_job_pending": (C.c_int64, [C.c_void_p]),
    "sh_job_depth": (C.c_int, [C.c_void_p]),
    "sh_glm_batch_dev_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "sh_set_lanes": (C.c_int, [C.c_void_p, C.c_int]),
    "sh_get_lanes": (C.c_int, [C.c_void_p]),
    "sh_format_records": (C.c_int64, [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.c_int64, C.POINTER(c_dp),
                                      C.c_int, c_dp, C.c_int64, C.c_int, c_u8p, C.c_void_p, C.c_void_p, C.c_int, c_u32p, C.POINTER(C.c_void_p)]),
    "sh_job_set_lineage": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_int]),
    "sh_job_set_patterns": (C.c_int, [C.c_void_p, C.c_int]),
    "sh_job_patterns": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "sh_job_set_samples": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]),
    "sh_job_run_packed": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p]),
    "sh_host_register": (C.c_int, [C.c_void_p, C.c_int64, C.c_int]),
    "sh_host_unregister": (C.c_int, [C.c_void_p]),
    "sh_host_cpus": (C.c_int, []),
    "sh_set_host_threads": (None, [C.c_int]),
    "sh_set_host_streams": (None, [C.c_int]),
    "sh_host_pool_workers": (C.c_int, []),
    "sh_host_cpu_seconds": (C.c_int, [C.c_char_p, C.c_int]),
    "sh_format_concurrency_max": (C.c_int, [C.c_int]),
}

_lib = None


class SeerHipError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libseerhip error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load libseerhip.so and bind every declared symbol (no device needed for this)."""
    global _lib
    if _lib is None:
        # torch first: libseerhip shares torch's HIP runtime (one libamdhip64 per process), which is what lets the engine
        # run on torch's streams and on torch-allocated HBM.  Loading our library before torch makes torch's later
        # runtime initialisation fail ("no ROCm-capable device").
        # TORCH_FIRST = False (the command line sets it when its options cannot reach torch: anything but a kinship decomposition on the
        # GPU): skips the 0.6 s import; importing torch afterwards in the same process is then an error the HIP runtime reports itself.
        if TORCH_FIRST:
            import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise ImportError("libseerhip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`. "
                              "There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.sh_abi_version() != 2:
            raise ImportError("libseerhip ABI version mismatch")
        _lib = lib
    return _lib


def host_cpu_seconds():
    """{stage: thread CPU seconds spent in it so far} for the host stages of the library (csrc/host_pool.h)."""
    buf = C.create_string_buffer(1024)
    n = load().sh_host_cpu_seconds(buf, len(buf))
    if n < 0:
        return {}
    return {k: float(v) for k, v in (item.split("=") for item in buf.value.decode().split(","))}


def check(rc):
    if rc != SH_OK:
        raise SeerHipError(rc, load().sh_last_error().decode())
    return rc
