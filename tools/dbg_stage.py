#!/usr/bin/env python3
"""one row of an end-to-end fixture through the ORACLE's Richardson / Wynn / de Hoog fed with the device's samples of
either flavour: separates 'the samples differ' from 'the device's own series acceleration / inversion reacts'.
usage: dbg_stage.py <deck> <radius index> <row>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from golden_util import load_e2e
from unconfined_amd import engine
import oracle_lib
O = oracle_lib.Oracle()
Oq = oracle_lib.Oracle(quad=True)
name, ir, row = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
e2e = load_e2e(name)
dk, P, D, t, tDv, rDv, svv, zD, zl = T._grid(O, name, ir, e2e)
tD, rD, sv = float(tDv[row]), float(rDv[row]), int(svv[row])
c = lambda x: x[..., 0] + 1j * x[..., 1]
p = O.pvalues(2 * tD, dk.M, dk.alpha, dk.tol); pc = c(p)
j0z = O.j0_zeros(D.nj0z)
arg = j0z[sv - 1] / rD
gx, gw = O.gauss_lobatto(dk.ord)
def pipeline(soln):
    tmp = []
    for j in range(1, dk.R + 1):
        w, a = O.tanh_sinh(dk.k - dk.R + j, arg)
        f = np.stack([c(soln(ai)) for ai in a])
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
            yy = np.array([[q[z, i].real, q[z, i].imag] for q in tmp])
            e = O.extrap(hv, yy); fin[z, i] = e[0] + 1j * e[1]
            ser = np.stack([areas[:, z, i].real, areas[:, z, i].imag], axis=1)
            a_, stt = O.wynn(ser); inf[z, i] = a_[0] + 1j * a_[1]
    tot = fin + inf
    ri = lambda v: np.stack([v.real, v.imag], axis=1)
    h = [O.dehoog(dk.M, dk.alpha, dk.tol, tD, 2 * tD, ri(tot[z])) for z in range(nz)]
    dh = [O.dehoog(dk.M, dk.alpha, dk.tol, tD, 2 * tD, ri(tot[z] * pc)) * tD for z in range(nz)]
    hq = [Oq.dehoog(dk.M, dk.alpha, dk.tol, tD, 2 * tD, ri(tot[z])) for z in range(nz)]
    dhq = [Oq.dehoog(dk.M, dk.alpha, dk.tol, tD, 2 * tD, ri(tot[z] * pc)) * tD for z in range(nz)]
    return np.array(h), np.array(dh), tot, np.array(hq), np.array(dhq)
res = {}
for mode in ("fast", "faithful"):
    pl = engine.Plan(P, mode=mode)
    res[mode] = pipeline(lambda a: pl.lap_hank_soln([a], rD, p, zD, zl)[0])
    hd, dd = pl.drawdown([tD], [rD], [sv], zD, zl)
    print(f"{mode:9s} device end to end: h {hd[0, 0]:.16e} dh {dd[0, 0]:.16e}")
res["oracle"] = pipeline(lambda a: O.soln(P, D, a, rD, p, zD, zl))
ho, dho = O.batch(P, np.array([tD]), np.array([rD]), np.array([sv], np.int32), zD, zl)
print(f"oracle    end to end       : h {ho[0, 0]:.16e} dh {dho[0, 0]:.16e}")
for k, v in res.items():
    print(f"{k:9s} samples -> oracle's Richardson/Wynn/de Hoog: h {v[0][0]:.16e} dh {v[1][0]:.16e}")
    print(f"{k:9s} samples -> oracle's Richardson/Wynn, de Hoog in binary128 on the same binary64 values: h {v[3][0]:.16e} dh {v[4][0]:.16e}")
ref = res["oracle"][2][0]
for k in ("fast", "faithful"):
    d = np.abs(res[k][2][0] - ref) / np.abs(ref)
    print(f"{k:9s} Laplace-space values against the oracle's: max rel diff {d.max():.2e} at m = {int(d.argmax())}; median {np.median(d):.2e}")
