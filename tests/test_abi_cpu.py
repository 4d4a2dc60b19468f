"""CPU-side checks of the C-ABI boundary: the library builds, loads and exports every declared symbol; without a GPU it
refuses to create a context (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from pyseer_amd import _abi
from pyseer_amd.packing import pack_variants, unpack_variants, row_bytes_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "seerhip.h")).read()
    declared = set(re.findall(r"\b(sh_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("sh_ctx")
    assert declared == set(_abi.SIGNATURES), (declared ^ set(_abi.SIGNATURES))
    lib = _abi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sh_abi_version() == 2


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _abi.load()
    assert lib.sh_device_count() == 0
    assert not lib.sh_create(0, 100)
    assert b"no HIP device" in lib.sh_last_error()
    from pyseer_amd.engine import Engine
    with pytest.raises(_abi.SeerHipError):
        Engine(100)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pyseer_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.lower(), "%s mentions the oracle" % f


def test_pack_roundtrip():
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 63, 64, 65, 100, 1000):
        K = (rng.random((5, n)) < 0.4).astype(np.uint8)
        b = pack_variants(K)
        assert b.shape == (5, row_bytes_for(n)) and b.shape[1] % 8 == 0
        assert (unpack_variants(b, n) == K).all()
        # LSB-first: sample i -> bit i&7 of byte i>>3
        i = n - 1
        assert ((b[:, i >> 3] >> (i & 7)) & 1 == K[:, i]).all()
