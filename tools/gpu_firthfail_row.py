import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyseer_amd.engine import Engine, pack_variants
d = np.load("tests/golden/glm_exit_firthfail_N300_q3_s9201.npz")
for mode, env in (("default", {}), ("literal", {"SEERHIP_FIRTH_LITERAL": "1"}), ("strict", {"SEERHIP_FIRTH_STRICT": "1"})):
    os.environ.update(env)
    e = Engine(int(d["N"])); e.set_af_filter(0.01, 0.99)
    e.glm_setup(d["y"], d["m"], False, float(d["null_llf"]), float(d["null_firth"]), 1.0, 1.0)
    r = e.glm_batch(pack_variants(d["K"])); e.close()
    for k in env: del os.environ[k]
    print(mode, "flags", (r["flags"] & 0x1FF).tolist(), "row 4:", r["pvalue"][4], r["kbeta"][4], r["bse"][4], r["intercept"][4])
print("reference row 4:", d["main"][4], "in other orders:", d["perm_main"][4, 0])
