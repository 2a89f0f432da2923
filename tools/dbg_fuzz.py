#!/usr/bin/env python3
"""replay flagged fuzz sets (tests/golden/fuzz_flagged_r02.json, or the fuzz_hunt.py result named by UCF_FUZZ_JSON):
sample-level errors of both flavours against binary128 along the abscissa range of the worst point.
usage: tools/dbg_fuzz.py seed set [seed set ...]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck
from unconfined_amd import engine
from unconfined_amd.abi import params_from_deck
import oracle_lib
O, Oq = oracle_lib.Oracle(), oracle_lib.Oracle(quad=True)
d = json.load(open(os.environ.get("UCF_FUZZ_JSON", os.path.join(ROOT, "tests", "golden", "fuzz_flagged_r02.json"))))
want = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
for r in d["flagged"]:
    if (r["seed"], r["set"]) not in want:
        continue
    dk = load_deck(r["base"])[0].replace(**r["change"])
    P = params_from_deck(dk)
    D = O.nondim(P)
    zD = np.array(r["zD"]); zl = np.array(r["zLay"], np.int32)
    pt = max(r["points"], key=lambda p: np.nanmax(np.abs(np.array(p["fast"]) - np.array(p["binary128"])) / np.maximum(np.abs(np.array(p["binary128"])), 1e-300)))
    tD, rD, sv = pt["tD"], pt["rD"], pt["sv"]
    print(f"== seed {r['seed']} set {r['set']} {r['base']} kappa {dk.kappa:.4g} l {dk.l:.4g} d {dk.d:.4g} b {dk.b} beta {dk.beta:.3g} zD {zD} lay {zl} tD {tD:.4g} rD {rD:.4g}")
    print("   dD", D.dD, "lD", D.lD, "fast", pt["fast"], "faithful", pt["faithful"], "truth", pt["binary128"])
    pf, pg = engine.Plan(P, mode="fast"), engine.Plan(P, mode="faithful")
    j0z = pf.j0z()
    p = O.pvalues(2 * tD, dk.M, dk.alpha, dk.tol)
    amax = j0z[sv - 1 + dk.nacc] / rD
    for a in list(np.linspace(0.02, 1.0, 6) * j0z[sv - 1] / rD) + list(np.linspace(j0z[sv - 1] / rD, amax, 10)):
        ff = pf.lap_hank_soln([a], rD, p, zD, zl)[0]; fg = pg.lap_hank_soln([a], rD, p, zD, zl)[0]
        ft = Oq.soln(P, D, a, rD, p, zD, zl); fo = O.soln(P, D, a, rD, p, zD, zl)
        c = lambda x: x[..., 0] + 1j * x[..., 1]
        zt = c(ft)
        err = lambda x: [float(np.linalg.norm((c(x) - zt)[z]) / max(np.linalg.norm(zt[z]), 1e-300)) for z in range(len(zD))]
        eta = np.sqrt((p[0, 0] + a * a) / dk.kappa)
        print(f"   a {a:9.3f} Re(eta0) {eta:8.2f} |f| {[float(np.linalg.norm(zt[z])) for z in range(len(zD))]} err fast {['%.1e' % e for e in err(ff)]} faithful {['%.1e' % e for e in err(fg)]} oracle {['%.1e' % e for e in err(fo)]}")
