import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pyseer_amd.engine import Engine
d = np.load(os.path.join(ROOT, "tests/golden/n5000_cap35.npz"))
e = Engine(5000); e.set_af_filter(0.01, 0.99)
e.glm_setup(d["y"].astype(float), d["W"], False, float(d["null_llf"]), float(d["null_firth"]), 1.0, 1.0)
r = e.glm_batch(np.ascontiguousarray(d["bits"])); e.close()
np.set_printoptions(linewidth=200, precision=8)
for v in range(d["bits"].shape[0]):
    print(v, "its", d["newton_iterations"][v], "ref notes", d["notes"][v], "hip flags", hex(r["flags"][v]))
    print("   ref", d["main"][v], d["betas"][v])
    print("   hip", np.array([r[k][v] for k in ("prep", "pvalue", "kbeta", "bse", "intercept")]), r["betas"][v])
