import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck, load_e2e
from unconfined_amd import engine
import oracle_lib
oracle = oracle_lib.Oracle()
name = "c4_malama_partpen"
dk, ts, P = load_deck(name); e2e = load_e2e(name)
D = oracle.nondim(P)
t = oracle.logspace(ts.min_log, ts.max_log, ts.n); tD = t / D.Tc
sv = oracle.split_vector(list(dk.j0s), tD)
zz = oracle.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd); zD = zz / D.Lc; zl = oracle.zlay(D, zD)
rD = np.full_like(tD, float(e2e["radii"][0]) / D.Lc)
print("rD", rD[0], "dD", D.dD, "lD", D.lD, "zD", zD, "zlay", zl, "kappa", dk.kappa)
res = {}
for mode in ("fast", "faithful"):
    plan = engine.Plan(P, mode=mode)
    res[mode] = plan.drawdown(tD, rD, sv, zD, zl)
ho, dho = oracle.batch(P, tD, rD, sv, zD, zl)
ref = e2e["O2_r0"]
sc = D.Hc if not dk.dimless else 1.0
for lab, k in (("h", 0), ("dh", 1)):
    f, g, o = res["fast"][k][:, 0] * sc, res["faithful"][k][:, 0] * sc, (ho, dho)[k][:, 0] * sc
    r = ref[:, 1 + k]
    den = np.maximum(np.abs(r), 1e-3)
    ef, eg, eo = np.abs(f - r) / den, np.abs(g - r) / den, np.abs(o - r) / den
    i = int(np.argmax(ef))
    print(lab, "worst row", i, "tD", tD[i], "fast", ef[i], "faithful", eg[i], "oracle", eo[i], "| max fast", ef.max(), "faithful", eg.max(), "oracle", eo.max())
