#!/usr/bin/env python3
"""throughput of the parameter-batched entry point (fitting / inversion): NP parameter sets x NPTS
observation points per call.  usage: tools/bench_multi.py [nplans] [npts]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck
from unconfined_amd import engine
from unconfined_amd.abi import params_from_deck

npl = int(sys.argv[1]) if len(sys.argv) > 1 else 64
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 240
dk, ts, P0 = load_deck("neuman74_partpen")
rng = np.random.default_rng(1)
plans = []
t0 = time.time()
for i in range(npl):
    d = dk.replace(Kr=dk.Kr * rng.uniform(0.5, 2.0), Sy=dk.Sy * rng.uniform(0.5, 1.5), kappa=dk.kappa * rng.uniform(0.5, 2.0),
                   Ss=dk.Ss * rng.uniform(0.5, 2.0))
    plans.append(engine.Plan(params_from_deck(d), mode="fast"))
t_plans = time.time() - t0
t0 = time.time()
for i, pl in enumerate(plans):       # the same sets again through ucf_plan_update
    pl.update(pl.params)
t_upd = time.time() - t0
t = 10.0 ** rng.uniform(-1, 4, npts); r = rng.choice([16.0, 30.0, 85.1, 150.0], npts); z = np.array([145.7])
engine.drawdown_multi(plans, t, r, z)       # warm-up (workspaces)
best = 1e9
for rep in range(3):
    t0 = time.time(); h, dh = engine.drawdown_multi(plans, t, r, z); best = min(best, time.time() - t0)
# one plan, all points in one call, for comparison
tD, rD = np.tile(t, npl) / plans[0].derived.Tc, np.tile(r, npl) / plans[0].derived.Lc
zD = z / plans[0].derived.Lc
plans[0].drawdown(tD, rD, plans[0].split_vector(tD), zD, plans[0].zlay(zD))
t0 = time.time(); plans[0].drawdown(tD, rD, plans[0].split_vector(tD), zD, plans[0].zlay(zD)); one = time.time() - t0
print(f"{npl} plans x {npts} points: multi {best * 1e3:.1f} ms = {npl * npts / best:.0f} points/s  "
      f"({best / npl * 1e6:.0f} us per plan); plan creation {t_plans / npl * 1e3:.2f} ms each, update {t_upd / npl * 1e6:.0f} us each; "
      f"same {npl * npts} points under one plan in one call: {one * 1e3:.1f} ms = {npl * npts / one:.0f} points/s")
