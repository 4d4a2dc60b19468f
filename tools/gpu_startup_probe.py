"""Where the command line's start-up goes at N = 5000 (after tools/gpu_e2e_c3.py in the same gpurun call: its inputs stay in /tmp/e2e_c3)."""
import os, sys, time
t00 = time.time()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
t = time.time(); import pandas as pd; print("import pandas %.2f" % (time.time() - t))
t = time.time(); import pyseer_amd.__main__ as M; print("import cli %.2f" % (time.time() - t))
from pyseer_amd.lmm import initialise_lmm
from pyseer_amd.input import load_phenotypes
from pyseer_amd.engine import Engine
d = "/tmp/e2e_c3"
t = time.time(); z = np.load(d + "/lmm.npz"); U = z["arr_0"]; S = z["arr_1"]; print("np.load %.2f" % (time.time() - t))
t = time.time(); p = load_phenotypes(d + "/pheno.tsv", None); print("phenotypes %.2f" % (time.time() - t))
cov = pd.DataFrame([])
t = time.time(); p2, lmm, h2 = initialise_lmm(p, cov, None, d + "/lmm.npz", None, lineage_samples=None, use_gpu=True, device=0); print("initialise_lmm %.2f" % (time.time() - t))
t = time.time(); e = Engine(len(p2)); print("Engine() %.2f" % (time.time() - t))
os.environ["SEERHIP_SETUP_DEBUG"] = "1"
t = time.time(); e.lmm_setup(lmm.U, lmm.S, lmm.Y, lmm.X, h2, False, 1.0, 1.0); print("lmm_setup %.2f" % (time.time() - t))
t = time.time(); e.lmm_setup(lmm.U, lmm.S, lmm.Y, lmm.X, h2, False, 1.0, 1.0); print("lmm_setup again %.2f" % (time.time() - t))
t = time.time(); e.close(); print("close %.2f" % (time.time() - t))
print("total %.2f" % (time.time() - t00))
