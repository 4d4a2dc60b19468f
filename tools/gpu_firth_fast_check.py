"""The one-pass Firth iteration (firth_fast.hip) against the two-pass rounds (SEERHIP_ROUTE firth_fast=0) on the C4 workload: same flags, statistics to
1e-6, and the time of a batch each way.  Also the reference's own answers on the 106 rows of tests/golden/n5000_firth.npz."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pyseer_amd.engine import Engine, row_bytes_for

N, q, V = 5000, int(os.environ.get("Q", 10)), int(os.environ.get("V", 262144))
y, W, nl, nf = bench.synth_glm_inputs(N, q)
dev = torch.device("cuda", 0)
bits = bench.synth_bits(V, N, row_bytes_for(N), 4242, dev)
res = {}
outs = {}
for mode, env in (("two_pass", {"SEERHIP_ROUTE": "firth_fast=0"}), ("one_pass", {})):
    for k, v_ in env.items():
        os.environ[k] = v_
    e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, False, nl, nf, force_firth=True)
    e.glm_batch_dev(bits); torch.cuda.synchronize()
    t0 = time.time(); o, f = e.glm_batch_dev(bits); torch.cuda.synchronize(); dt = time.time() - t0
    e.close()
    for k in env:
        del os.environ[k]
    outs[mode] = (o.cpu().numpy(), f.cpu().numpy().astype(np.uint32))
    res[mode] = {"ms_per_batch": dt * 1e3, "variants_per_s": V / dt}
a, fa = outs["two_pass"]; b, fb = outs["one_pass"]
res["flags_equal"] = bool(np.array_equal(fa, fb)); res["rows_flags_differ"] = int((fa != fb).sum())
ok = np.isfinite(a[2]) & np.isfinite(b[2])
res["both_finite"] = int(ok.sum()); res["finite_only_one"] = int((np.isfinite(a[2]) != np.isfinite(b[2])).sum())
for name, r in (("pvalue", 1), ("kbeta", 2), ("bse", 3), ("intercept", 4), ("beta_z0", 5)):
    d = np.abs(a[r][ok] - b[r][ok]); rel = d / np.maximum(np.abs(a[r][ok]), 1e-300)
    res[name] = {"max_abs": float(d.max()), "max_rel": float(rel.max()), "rows_rel_over_1e-6": int((rel > 1e-6).sum())}
print(json.dumps(res, indent=1))
# the reference's own numbers
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(root, "tests", "golden", "n5000_firth.npz"))
if q == 10:
    e = Engine(N); e.set_af_filter(0.01, 0.99)
    e.glm_setup(d["y"].astype(float), d["W"], False, float(d["null_llf"]), float(d["null_firth"]), force_firth=True)
    r = e.glm_batch(np.ascontiguousarray(d["bits"])); e.close()
    fm = d["firth_main"]
    print("vs reference (106 rows): firth-fail %d, max rel dkbeta %.3g, dbse %.3g, max abs dintercept %.3g" % (
        int(((r["flags"] >> 6) & 1).sum()), float(np.max(np.abs(r["kbeta"] - fm[:, 1]) / np.abs(fm[:, 1]))),
        float(np.max(np.abs(r["bse"] - fm[:, 2]) / fm[:, 2])), float(np.max(np.abs(r["intercept"] - fm[:, 0])))))
o_ = os.path.join(os.environ.get("GRAFT_REPO_ROOT", root), "gpurun_out", "r04"); os.makedirs(o_, exist_ok=True)
json.dump(res, open(os.path.join(o_, "firth_fast_check_q%d.json" % q), "w"), indent=1)
