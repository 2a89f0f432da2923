#!/usr/bin/env python3
"""contour-style call: many depths per (t, r) point; times the grid entry point (depths are walked in chunks)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck
from unconfined_amd import engine
name = sys.argv[1] if len(sys.argv) > 1 else "c2_neuman74_fullpen"
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 21
dk, ts, P = load_deck(name)
plan = engine.Plan(P, mode="fast")
tD = 10.0 ** np.linspace(-1, 4, 128); sv = plan.split_vector(tD)
rD = 10.0 ** np.linspace(-1, 1, 64)
zD = np.linspace(0.03, 0.97, nz); zl = plan.zlay(zD)
plan.drawdown_grid(tD, sv, rD, zD, zl)
best = 1e9
for _ in range(3):
    t0 = time.time(); h, dh = plan.drawdown_grid(tD, sv, rD, zD, zl); best = min(best, time.time() - t0)
print(f"{name}: {len(tD)} x {len(rD)} x {nz} depths: {best * 1e3:.1f} ms = {len(tD) * len(rD) * nz / best:.0f} (t,r,z) values/s  UCF_Z_CHUNK={os.environ.get('UCF_Z_CHUNK', 'auto')}")
