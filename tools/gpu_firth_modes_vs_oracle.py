"""Which Firth step-halving mode agrees with the reference restatement (the CPU oracle, oracle/seer_oracle.c orc_fit_firth, both test knobs 0)
row by row?  C4 workload (N = 5000, 10 covariates, every variant through fit_firth), V variants:
  noise     (default) an increase within 4 ulp of F is not an increase, steps below 1e-10 are accepted
  literal   SEERHIP_ROUTE firth_literal=1: the reference's `F(new) > F(old)`, F as the round kernels evaluate it
  strict    SEERHIP_ROUTE firth_strict=1: literal rule, one log per sample, the reference's start vector
Counts firth-fail rows of each and of the oracle, their overlaps, and the largest |kbeta - oracle| over rows both fit.  Writes gpurun_out/r03/."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pyseer_amd.engine import Engine, row_bytes_for
from oracle import oracle as orc

N, q, V = 5000, 10, int(os.environ.get("V", 24576))
y, W, nl, nf = bench.synth_glm_inputs(N, q)
dev = torch.device("cuda", 0)
bits = bench.synth_bits(V, N, row_bytes_for(N), 4242, dev)
rows = bench.unpack_rows(bits.cpu().numpy(), N)
af = rows.mean(axis=1); inwin = (af >= 0.01) & (af <= 0.99)
orc.set_threads(bench.effective_cpus())
t0 = time.time(); w = orc.firth_batch(y, rows, W); t_or = time.time() - t0
ofail = (w["status"] != 0) & inwin
res = {"workload": "C4: N=%d, q=%d, %d variants (%d inside the AF window), force_firth" % (N, q, V, int(inwin.sum())),
       "oracle_firth_fail": int(ofail.sum()), "oracle_seconds": t_or}
out = {}
for mode, env in (("noise", {}), ("literal", {"SEERHIP_ROUTE": "firth_literal=1"}), ("strict", {"SEERHIP_ROUTE": "firth_strict=1"})):
    for k, v_ in env.items():
        os.environ[k] = v_
    e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, False, nl, nf, force_firth=True)
    e.glm_batch_dev(bits); torch.cuda.synchronize()
    t0 = time.time(); o, f = e.glm_batch_dev(bits); torch.cuda.synchronize(); dt = time.time() - t0
    e.close()
    for k in env:
        del os.environ[k]
    o = o.cpu().numpy(); f = f.cpu().numpy().astype(np.uint32)
    gfail = (((f >> 6) & 1) == 1) & inwin
    both = inwin & ~gfail & ~ofail
    dk = np.abs(o[2][both] - w["kbeta"][both]); rel = dk / np.maximum(np.abs(w["kbeta"][both]), 1e-300)
    res[mode] = {"firth_fail": int(gfail.sum()), "fail_in_both": int((gfail & ofail).sum()), "fail_only_here": int((gfail & ~ofail).sum()),
                 "fail_only_oracle": int((~gfail & ofail).sum()), "rows_with_a_flag_difference": int((gfail != ofail).sum()),
                 "max_abs_dkbeta": float(dk.max()), "rows_rel_dkbeta_over_1e-6": int((rel > 1e-6).sum()), "rows_abs_dkbeta_over_1e-7": int((dk > 1e-7).sum()),
                 "variants_per_s": V / dt}
print(json.dumps(res, indent=1))
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r03"); os.makedirs(d, exist_ok=True)
json.dump(res, open(os.path.join(d, "firth_modes_vs_reference_restatement.json"), "w"), indent=1)
