import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck
from unconfined_amd import engine
from unconfined_amd.abi import params_from_deck
import oracle_lib
oracle, oq = oracle_lib.Oracle(), oracle_lib.Oracle(quad=True)
dk, ts, P0 = load_deck("neuman74_partpen")
dk = dk.replace(kappa=0.16)
P = params_from_deck(dk)
D = oracle.nondim(P)
zD = np.array([0.93, 0.80, 0.30]); rD = 0.117; tD = 98.0
pf, pg = engine.Plan(P, mode="fast"), engine.Plan(P, mode="faithful")
zl = pf.zlay(zD)
print("dD", D.dD, "lD", D.lD, "zlay", zl)
p = oracle.pvalues(2 * tD, dk.M, dk.alpha, dk.tol)
for a in (2.0, 20.0, 60.0, 120.0, 200.0, 250.0):
    ff = pf.lap_hank_soln([a], rD, p, zD, zl)[0]
    fg = pg.lap_hank_soln([a], rD, p, zD, zl)[0]
    fr = oracle.soln(P, D, a, rD, p, zD, zl)
    ft = oq.soln(P, D, a, rD, p, zD, zl)
    c = lambda x: x[..., 0] + 1j * x[..., 1]
    zf, zg, zr, zt = c(ff), c(fg), c(fr), c(ft)
    eta = np.sqrt((a * a + p[0, 0]) / dk.kappa)
    for iz in range(3):
        n = np.linalg.norm(zt[iz])
        print(f"a={a:6.1f} eta~{eta:6.1f} zD={zD[iz]:.2f} lay={zl[iz]} |f|={n:.2e} err/|f|: fast {np.linalg.norm(zf[iz]-zt[iz])/n:.2e} faithful {np.linalg.norm(zg[iz]-zt[iz])/n:.2e} oracle {np.linalg.norm(zr[iz]-zt[iz])/n:.2e}")
