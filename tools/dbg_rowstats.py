"""in-band rule counters (ucf_stats / the oracle's) and dh of both flavours for a window of rows of an end-to-end fixture.
usage: dbg_rowstats.py [deck] [radius index] [first row] [last row]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from golden_util import load_e2e
from unconfined_amd import engine
import oracle_lib
oracle = oracle_lib.Oracle()
name = sys.argv[1] if len(sys.argv) > 1 else "c2_neuman74_fullpen"; e2e=load_e2e(name)
ir = int(sys.argv[2]) if len(sys.argv) > 2 else 2
r0 = int(sys.argv[3]) if len(sys.argv) > 3 else 166
r1 = int(sys.argv[4]) if len(sys.argv) > 4 else 170
dk,P,D,t,tD,rD,sv,zD,zl = T._grid(oracle, name, ir, e2e)
for r in range(r0, r1 + 1):
    out=[]
    for m in ("fast","faithful"):
        h,dh,st = engine.Plan(P,mode=m).drawdown(tD[r:r+1], rD[r:r+1], sv[r:r+1], zD, zl, with_stats=True)
        out.append((m, float(dh[0,0]), {k:v for k,v in st.items() if v}))
    ho,dho,so = oracle.batch_with_stats(P, tD[r:r+1], rD[r:r+1], sv[r:r+1], zD, zl)
    print(r, out, 'oracle', float(dho[0,0]), {k:v for k,v in so.items() if v})
