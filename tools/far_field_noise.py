#!/usr/bin/env python3
"""error statistics of the fast flavour against the binary128 evaluation at far-field points (rD 3 ... 10, where the theis
term of the Hankel integral cancels to a small fraction of its intervals) and at near-field points, for the library named by
UCF_LIB_PATH -- used to compare evaluator variants statistically, not by their unluckiest point.
usage: far_field_noise.py [deck] [npts]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck
from unconfined_amd import engine
import oracle_lib
O, Oq = oracle_lib.Oracle(), oracle_lib.Oracle(quad=True)
name = sys.argv[1] if len(sys.argv) > 1 else "c2_neuman74_fullpen"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 192
dk, ts, P = load_deck(name)
D = O.nondim(P)
rng = np.random.default_rng(11)
zD = np.array([0.2, 0.55, 0.9])
for label, rlo, rhi in (("far  (rD 3..10)", 3.0, 10.0), ("near (rD 0.1..1)", 0.1, 1.0)):
    tD = 10.0 ** rng.uniform(-1.0, 3.0, n); rD = rng.uniform(rlo, rhi, n)
    res = {}
    for mode in ("fast", "faithful"):
        pl = engine.Plan(P, mode=mode); zl = pl.zlay(zD); sv = pl.split_vector(tD)
        res[mode] = pl.drawdown(tD, rD, sv, zD, zl)[0]
    ho, _ = O.batch(P, tD, rD, sv, zD, zl)
    ht, _ = Oq.batch(P, tD, rD, sv, zD, zl, threads=8)
    sc = np.maximum(np.abs(ht), 1e-6)
    line = [label]
    for k, v in (("fast", res["fast"]), ("faithful", res["faithful"]), ("oracle", ho)):
        e = (np.abs(v - ht) / sc).ravel()
        line.append(f"{k}: median {np.median(e):.2e} p90 {np.percentile(e, 90):.2e} p99 {np.percentile(e, 99):.2e} max {e.max():.2e}")
    print(" | ".join(line))
