#!/usr/bin/env python3
"""one flagged point through the oracle's own Richardson / Wynn / de Hoog, fed with the DEVICE's samples of either
flavour: separates 'the samples differ' from 'the series acceleration reacts to them'"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck
from unconfined_amd import engine
from unconfined_amd.abi import params_from_deck
import oracle_lib
O, Oq = oracle_lib.Oracle(), oracle_lib.Oracle(quad=True)
d = json.load(open(os.path.join(ROOT, "tests", "golden", "fuzz_flagged_r02.json")))
seed, st = int(sys.argv[1]), int(sys.argv[2])
r = next(x for x in d["flagged"] if (x["seed"], x["set"]) == (seed, st))
dk = load_deck(r["base"])[0].replace(**r["change"])
P = params_from_deck(dk); D = O.nondim(P)
zD = np.array(r["zD"]); zl = np.array(r["zLay"], np.int32)
pt = max(r["points"], key=lambda p: np.nanmax(np.abs(np.array(p["fast"]) - np.array(p["binary128"])) / np.maximum(np.abs(np.array(p["binary128"])), 1e-300)))
tD, rD, sv = pt["tD"], pt["rD"], pt["sv"]
c = lambda x: x[..., 0] + 1j * x[..., 1]
p = O.pvalues(2 * tD, dk.M, dk.alpha, dk.tol); pc = c(p)
j0z = O.j0_zeros(D.nj0z)
arg = j0z[sv - 1] / rD
gx, gw = O.gauss_lobatto(dk.ord)
def pipeline(soln):
    # finite part: Richardson over tanh-sinh levels
    out = {}
    tmp = []
    for j in range(1, dk.R + 1):
        kv = dk.k - dk.R + j
        w, a = O.tanh_sinh(kv, arg)
        f = np.stack([c(soln(ai)) for ai in a])              # [n][nz][np]
        tmp.append(arg / 2.0 * np.tensordot(w, f, axes=(0, 0)))
    hv = np.array([4.0 / 2 ** (dk.k - dk.R + j) for j in range(1, dk.R + 1)])
    nz, npp = len(zD), len(pc)
    fin = np.zeros((nz, npp), complex); inf = np.zeros((nz, npp), complex)
    areas = np.zeros((dk.nacc, nz, npp), complex)
    for jj in range(dk.nacc):
        lob, hib = j0z[sv + jj - 1] / rD, j0z[sv + jj] / rD
        y = ((hib - lob) * gx + (hib + lob)) / 2.0
        f = np.stack([c(soln(yi)) for yi in y])
        areas[jj] = (hib - lob) / 2.0 * np.tensordot(gw, f, axes=(0, 0))
    for z in range(nz):
        for i in range(npp):
            yy = np.array([[t[z, i].real, t[z, i].imag] for t in tmp])
            e = O.extrap(hv, yy); fin[z, i] = e[0] + 1j * e[1]
            ser = np.stack([areas[:, z, i].real, areas[:, z, i].imag], axis=1)
            a_, stt = O.wynn(ser); inf[z, i] = a_[0] + 1j * a_[1]
    tot = fin + inf
    h = [O.dehoog(dk.M, dk.alpha, dk.tol, tD, 2 * tD, np.stack([tot[z].real, tot[z].imag], axis=1)) for z in range(nz)]
    return h, fin, inf, areas
res = {}
for mode in ("fast", "faithful"):
    pl = engine.Plan(P, mode=mode)
    res[mode] = pipeline(lambda a: pl.lap_hank_soln([a], rD, p, zD, zl)[0])
res["oracle"] = pipeline(lambda a: O.soln(P, D, a, rD, p, zD, zl))
print("point", tD, rD, "device e2e fast", pt["fast"], "faithful", pt["faithful"], "truth", pt["binary128"])
for k, v in res.items():
    print(k, "host-pipeline h", v[0])
z = len(zD) - 1
for i in (0, 1, 5, 20):
    print("p index", i)
    for k, v in res.items():
        print("  ", k, "finint", v[1][z, i], "infint", v[2][z, i])
        print("      |areas|", ["%.2e" % abs(x) for x in v[3][:, z, i]])
