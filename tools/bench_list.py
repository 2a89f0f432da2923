#!/usr/bin/env python3
"""a long point list (lane = point layout) with nz depths: bench_list.py DECK NPTS NZ"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck
from unconfined_amd import engine
name = sys.argv[1] if len(sys.argv) > 1 else "neuman74_partpen"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
nz = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dk, ts, P = load_deck(name)
plan = engine.Plan(P, mode="fast")
rng = np.random.default_rng(1)
tD = 10.0 ** rng.uniform(-1, 4, n); rD = 10.0 ** rng.uniform(-1, 1, n); sv = plan.split_vector(tD)
zD = np.linspace(0.2, 0.8, nz); zl = plan.zlay(zD)
plan.drawdown(tD, rD, sv, zD, zl)
best = 1e9
for _ in range(3):
    t0 = time.time(); h, dh = plan.drawdown(tD, rD, sv, zD, zl); best = min(best, time.time() - t0)
print(f"{name}: list of {n} points x {nz} depths: {best * 1e3:.1f} ms = {n / best:.0f} points/s  UCF_NZC2={os.environ.get('UCF_NZC2', 'on')}")
