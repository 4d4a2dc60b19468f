"""The CPU baselines bench.py times beside the GPU are themselves checked: oracle/lmm_blas.py (the reference's BLAS formulation of
fit_lmm_block, pyseer/lmm.py:228-260 over fastlmm/lmm_cov.py:165-194, 597-838) against the reference's golden blocks and the C oracle,
and the multi-process runner (oracle/cpu_baseline_lmm.py, the `--cpu P` arrangement of pyseer/__main__.py:541-568)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_lmm_blas_matches_reference_goldens_and_c_oracle():
    from oracle.lmm_blas import LmmBlas
    from oracle import oracle as orc
    for name in ("lmm_N300_D3.npz", "lmm_N300_D1.npz", "lmm_N50_D1.npz"):
        d = np.load(os.path.join(G, name))
        L = LmmBlas(d["U"], d["S"], d["y"], d["covar"])
        O = orc.LmmOracle(d["U"], d["S"], d["y"], d["covar"])
        Kv = d["Kv"].astype(float)
        for key in d["h2_keys"]:
            h2 = float(str(key).split("_")[1])
            if abs(h2 - float(d["h2"])) < 1e-6:
                h2 = float(d["h2"])
            want = d["blk_" + str(key)]
            with np.errstate(all="ignore"):
                got = L.block(h2, Kv); ref = O.block(h2, Kv)
            for c, (g, r) in enumerate(zip(got, ref)):
                w = want[:, (0, 1, 2, 3)[c]]
                ok = np.isfinite(w) & np.isfinite(r) & (np.abs(w) > 1e-9) & (np.abs(want[:, 1]) > 1e-7)
                assert np.allclose(g[ok], w[ok], rtol=1e-7), (name, key, c)
                assert np.allclose(g[ok], r[ok], rtol=1e-9), (name, key, c)
    import pytest
    with pytest.raises(KeyError):
        L.block(1.0, Kv)


def test_cpu_baseline_runner(tmp_path):
    d = np.load(os.path.join(G, "lmm_N300_D3.npz"))
    f = str(tmp_path / "in.npz")
    np.savez(f, U=d["U"], S=d["S"], y=d["y"], C=d["covar"], h2=float(d["h2"]))
    env = dict(os.environ); env["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline_lmm", f, "2", "200", "2"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["variants"] == 2 * 2 * 200 and r["procs"] == 2 and r["seconds"] > 0
