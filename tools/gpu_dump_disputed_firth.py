"""Dump the forced-Firth rows of the C4 workload on which the three step-halving modes and the CPU restatement disagree, so that
the REAL reference (pyseer.model.fit_firth, importable only in the build container) can be run on exactly those rows
(tests/golden/make_n5000_golden.py).  Same workload and seed as tools/gpu_firth_modes_vs_oracle.py (N = 5000, q = 10, seed 4242).
A row is disputed when any of {noise (default), literal, strict, restatement} differs from another in its firth-fail flag, or two
of them that both fit differ by more than 1e-7 in kbeta.  Also keeps 24 undisputed rows (every 1000th) as controls.
Writes gpurun_out/r04/disputed_firth_rows.npz: packed bits, row indices, per-mode (kbeta, bse, intercept, pvalue, flags)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pyseer_amd.engine import Engine, row_bytes_for
from oracle import oracle as orc

N, q, V = 5000, 10, int(os.environ.get("V", 24576))
y, W, nl, nf = bench.synth_glm_inputs(N, q)
dev = torch.device("cuda", 0)
bits = bench.synth_bits(V, N, row_bytes_for(N), 4242, dev)
hb = bits.cpu().numpy()
rows = bench.unpack_rows(hb, N)
af = rows.mean(axis=1); inwin = (af >= 0.01) & (af <= 0.99)
orc.set_threads(bench.effective_cpus())
w = orc.firth_batch(y, rows, W)
res = {"restatement": dict(kbeta=w["kbeta"], bse=w["bse"], intercept=w["intercept"], fail=(w["status"] != 0) & inwin)}
for mode, env in (("noise", {}), ("literal", {"SEERHIP_ROUTE": "firth_literal=1"}), ("strict", {"SEERHIP_ROUTE": "firth_strict=1"})):
    for k, v_ in env.items():
        os.environ[k] = v_
    e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, False, nl, nf, force_firth=True)
    o, f = e.glm_batch_dev(bits); torch.cuda.synchronize()
    e.close()
    for k in env:
        del os.environ[k]
    o = o.cpu().numpy(); f = f.cpu().numpy().astype(np.uint32)
    res[mode] = dict(kbeta=o[2], bse=o[3], intercept=o[4], pvalue=o[1], flags=f, fail=(((f >> 6) & 1) == 1) & inwin)
names = ["restatement", "noise", "literal", "strict"]
disp = np.zeros(V, dtype=bool)
for i, a in enumerate(names):
    for b in names[i + 1:]:
        fa, fb = res[a]["fail"], res[b]["fail"]
        disp |= (fa != fb) & inwin
        both = inwin & ~fa & ~fb
        disp |= both & (np.abs(res[a]["kbeta"] - res[b]["kbeta"]) > 1e-7)
ctrl = np.zeros(V, dtype=bool); ctrl[::1000] = True; ctrl &= inwin & ~disp
keep = np.where(disp | ctrl)[0]
out = dict(N=N, q=q, seed_bits=4242, index=keep, bits=hb[keep], disputed=disp[keep], y=y, W=W, null_llf=nl, null_firth=nf)
for n_ in names:
    for k_, v_ in res[n_].items():
        out["%s_%s" % (n_, k_)] = np.asarray(v_)[keep]
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04"); os.makedirs(d, exist_ok=True)
np.savez_compressed(os.path.join(d, "disputed_firth_rows.npz"), **out)
print(json.dumps({"V": V, "in_window": int(inwin.sum()), "disputed": int(disp.sum()), "controls": int(ctrl.sum()),
                  "fails": {n_: int(res[n_]["fail"].sum()) for n_ in names}}))
