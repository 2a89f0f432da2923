"""one radius of an end-to-end fixture, a window of rows: fast / faithful / binary64 oracle / reference .out against the
binary128 evaluation.  usage: dbg_rows.py <deck> <radius index> <first row> <last row>"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from golden_util import load_e2e, rel_err
from unconfined_amd import engine
import oracle_lib
oracle, quad = oracle_lib.Oracle(), oracle_lib.Oracle(quad=True)
name, ir, r0, r1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
e2e = load_e2e(name)
dk, P, D, t, tD, rD, sv, zD, zl = T._grid(oracle, name, ir, e2e)
rows = np.arange(r0, r1 + 1)
ht, dht = quad.batch(P, tD[rows], rD[rows], sv[rows], zD, zl, threads=8)
ho, dho = oracle.batch(P, tD[rows], rD[rows], sv[rows], zD, zl)
res = {m: engine.Plan(P, mode=m).drawdown(tD[rows], rD[rows], sv[rows], zD, zl) for m in ("fast", "faithful")}
fl = 1e-3 / (1.0 if dk.dimless else D.Hc)
print("row  t        dh_truth       err: oracle   fast     faithful  | h: oracle   fast     faithful")
for i, r in enumerate(rows):
    e = lambda a, b: float(rel_err(a[i, 0], b[i, 0], fl))
    print(f"{r:4d} {t[r]:8.4g} {dht[i,0]:14.6e}   {e(dho,dht):.2e} {e(res['fast'][1],dht):.2e} {e(res['faithful'][1],dht):.2e}  |  "
          f"{e(ho,ht):.2e} {e(res['fast'][0],ht):.2e} {e(res['faithful'][0],ht):.2e}")
