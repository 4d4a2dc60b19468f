"""How often do the default Firth noise rules change an output?  C4 workload (N = 5000, 10 covariates, every variant through Firth) in the
default mode and with SEERHIP_ROUTE firth_strict=1 (the reference's literal step-halving test); writes a small JSON (copied to profiles/r02/)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pyseer_amd.engine import Engine, row_bytes_for

N, q, V = 5000, 10, int(os.environ.get("V", 1 << 17))
y, W, nl, nf = bench.synth_glm_inputs(N, q)
dev = torch.device("cuda", 0)
bits = bench.synth_bits(V, N, row_bytes_for(N), 4242, dev)
res = {}
for mode in ("default", "strict"):
    if mode == "strict":
        os.environ["SEERHIP_ROUTE"] = "firth_strict=1"
    e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, False, nl, nf, force_firth=True)
    e.glm_batch_dev(bits); torch.cuda.synchronize()
    t0 = time.time(); o, f = e.glm_batch_dev(bits); torch.cuda.synchronize(); dt = time.time() - t0
    res[mode] = (o.cpu().numpy(), f.cpu().numpy().astype(np.uint32), V / dt)
    e.close()
od, fd, rd = res["default"]; os_, fs, rs = res["strict"]
fit = (fd & 1) == 0
ffd, ffs = (fd >> 6) & 1, (fs >> 6) & 1
both = fit & (ffd == 0) & (ffs == 0)
dk = np.abs(od[2][both] - os_[2][both])
out = {"workload": "C4: N=%d, q=%d, %d variants, force_firth" % (N, q, V), "fitted": int(fit.sum()),
       "firth_fail_default": int(ffd[fit].sum()), "firth_fail_strict": int(ffs[fit].sum()),
       "rows_with_different_flags": int((fd != fs).sum()),
       "rows_kbeta_differs_more_than_1e-9": int((dk > 1e-9).sum()), "rows_kbeta_differs_more_than_1e-7": int((dk > 1e-7).sum()),
       "max_abs_kbeta_difference": float(dk.max()), "variants_per_s_default": rd, "variants_per_s_strict": rs}
print(json.dumps(out))
os.makedirs(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r02"), exist_ok=True)
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r02", "firth_strict_vs_default.json"), "w"), indent=1)
