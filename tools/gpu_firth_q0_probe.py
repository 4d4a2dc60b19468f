"""Development probe: forced-Firth results at q = 0 under the lean / plain information pass and several sample splits."""
import os, subprocess, sys, json
import numpy as np
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    N = int(sys.argv[1]); rng = np.random.default_rng(5)
    y = (rng.random(N) < 0.4).astype(float); W = np.zeros((N, 0)); K = (rng.random((40, N)) < rng.uniform(0.1, 0.9, 40)[:, None]).astype(np.uint8)
    e0 = np.zeros((0, 0)); null = fit_null(y, W, e0, False); nf = fit_null(y, W, e0, False, firth=True)
    e = Engine(N); e.glm_setup(y, W, False, null.llf, nf, force_firth=True); r = e.glm_batch(pack_variants(K)); e.close()
    print(json.dumps({"kbeta": [round(x, 9) for x in r["kbeta"][:8].tolist()]}))
else:
    for N in (320,):
        for env in ({"SEERHIP_FIRTH_LEAN": "0"}, {"SEERHIP_FIRTH_LEAN": "1"}, {"SEERHIP_FIRTH_LEAN": "2"}, {"SEERHIP_FIRTH_LEAN": "3"}):
            out = subprocess.run([sys.executable, __file__, str(N)], env=dict(os.environ, **env), capture_output=True, text=True)
            print(N, env, out.stdout.strip()[-400:], out.stderr.strip()[-300:])
