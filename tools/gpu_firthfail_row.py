"""The reference's own `firth-fail` rows (tests/golden/glm_exit_firthfail_*.npz: found by make_exit_golden.py --screened, every one a
rounding-noise failure that the reference does not repeat with the samples in another order) under the three step-halving modes of the HIP
path.  Prints per fixture and mode which of those rows come back `firth-fail`, and the largest relative distance of the fitted ones from
the reference's answer in another sample order.  -> gpurun_out/r04/firthfail_rows_modes.json"""
import glob, json, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from pyseer_amd.engine import Engine, pack_variants
res = {}
for path in sorted(glob.glob(os.path.join(root, "tests", "golden", "glm_exit_firthfail_*.npz"))):
    d = np.load(path); name = os.path.basename(path)[:-4]
    ff = np.where(d["notes"] & 0x40)[0]
    res[name] = {"N": int(d["N"]), "q": int(d["q"]), "reference_firth_fail_rows": ff.tolist(),
                 "reference_notes_in_three_other_sample_orders": d["perm_notes"][ff].tolist()}
    for mode, env in (("default", {}), ("literal", {"SEERHIP_ROUTE": "firth_literal=1"}), ("strict", {"SEERHIP_ROUTE": "firth_strict=1"})):
        os.environ.update(env)
        e = Engine(int(d["N"])); e.set_af_filter(0.01, 0.99)
        e.glm_setup(d["y"], d["m"], False, float(d["null_llf"]), float(d["null_firth"]), 1.0, 1.0)
        r = e.glm_batch(pack_variants(d["K"])); e.close()
        for k in env:
            del os.environ[k]
        fails = [int(v) for v in ff if (r["flags"][v] >> 6) & 1]
        worst = 0.0
        for v in ff:
            if (r["flags"][v] >> 6) & 1:
                continue
            j = [j for j in range(3) if not (d["perm_notes"][v, j] & 0x40)][0]
            pm = d["perm_main"][v, j]
            got = np.array([r["pvalue"][v], r["kbeta"][v], r["bse"][v], r["intercept"][v]])
            worst = max(worst, float(np.max(np.abs(got - pm[1:]) / np.abs(pm[1:]))))
        res[name][mode] = {"firth_fail_on": fails, "max_rel_distance_of_the_fitted_rows_from_the_reference_in_another_order": worst}
print(json.dumps(res, indent=1))
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", root), "gpurun_out", "r04"); os.makedirs(o, exist_ok=True)
json.dump(res, open(os.path.join(o, "firthfail_rows_modes.json"), "w"), indent=1)
