"""Dump the first rows of bench.py's C4 / C2N5000 workload (fixed_effects_line's first batch: synth_bits(2^18, 5000, rb, 4242)) so that the
reference itself can fit them in the build container (tests/golden/make_n5000_golden.py firth_random / glm_random): the rows are drawn by
torch's device generator, which the container (no GPU) cannot replay.  Also stores what the HIP path returns for them today."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from pyseer_amd.engine import Engine, row_bytes_for

N, q, NR = 5000, 10, int(os.environ.get("ROWS", 256))
dev = torch.device("cuda", 0)
rb = row_bytes_for(N)
y, W, nl, nf = bench.synth_glm_inputs(N, q)
bits = bench.synth_bits(1 << 18, N, rb, 4242, dev)
out = {"N": N, "q": q, "seed_bits": 4242, "y": y.astype(np.uint8), "W": W, "bits": bits[:NR].cpu().numpy(), "null_llf": nl, "null_firth": nf}
for name, force in (("firth", True), ("glm", False)):
    e = Engine(N, device=0); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, False, nl, nf, 1.0, 1.0, force_firth=force)
    r = e.glm_batch(out["bits"])
    for k, v in r.items():
        out["hip_%s_%s" % (name, k)] = v
    e.close()
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "r05"); os.makedirs(o, exist_ok=True)
np.savez_compressed(o + "/c4_random_rows.npz", **out)
print("dumped %d rows; carriers min/median/max %d/%d/%d" % (NR, *np.percentile(np.unpackbits(out["bits"], axis=1).sum(axis=1), [0, 50, 100])))
