import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_n5000_golden_gpu as T
d = np.load(os.path.join(T.G, "n5000_random.npz"))
ok = d["firth_ok"] == 1
fm = d["firth_main"]
K = np.unpackbits(d["bits"], axis=1, bitorder="little")[:, :5000]
for n in ("1", "2"):
    os.environ["SEERHIP_ROUTE"] = "firth_first32=" + n
    r = T._glm(d, d["bits"], force_firth=True)
    da = np.abs(r["kbeta"][ok] - fm[ok, 1]); rel = da / np.abs(fm[ok, 1])
    print("n32=%s: kbeta abs max %.3g median %.3g; rel max %.3g; rows rel > 1e-6: %d; bse rel max %.3g; intercept rel max %.3g" % (
        n, da.max(), np.median(da), rel.max(), int((rel > 1e-6).sum()), np.max(np.abs(r["bse"][ok] - fm[ok, 2]) / fm[ok, 2]),
        np.max(np.abs(r["intercept"][ok] - fm[ok, 0]) / np.abs(fm[ok, 0]))))
    idx = np.flatnonzero(ok)
    for i in np.argsort(-rel)[:5]:
        v = idx[i]
        print("   row %d carriers %d kbeta %.6g abs dev %.3g rel %.3g" % (v, int(K[v].sum()), fm[v, 1], da[i], rel[i]))
