#!/usr/bin/env python3
"""CPU only: how much de Hoog's inversion (oracle, binary64) of one row of an end-to-end fixture amplifies a relative
perturbation of a single Laplace-space value -- max over the 2M+1 values, for h and for dh.
usage: dbg_amplification.py <deck> <radius index> <first row> <last row>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck, load_e2e
import oracle_lib
O = oracle_lib.Oracle()
name, ir, r0, r1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
e2e = load_e2e(name)
dk, ts, P = load_deck(name)
D = O.nondim(P)
t = O.logspace(ts.min_log, ts.max_log, ts.n); tDv = t / D.Tc
svv = O.split_vector(list(dk.j0s), tDv)
zD = O.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd) / D.Lc; zl = O.zlay(D, zD)
rD = float(e2e["radii"][ir]) / D.Lc
c = lambda x: x[..., 0] + 1j * x[..., 1]
ri = lambda v: np.stack([v.real, v.imag], axis=1)
gx, gw = O.gauss_lobatto(dk.ord)
j0z = O.j0_zeros(D.nj0z)
for row in range(r0, r1 + 1):
    tD, sv = float(tDv[row]), int(svv[row])
    p = O.pvalues(2 * tD, dk.M, dk.alpha, dk.tol); pc = c(p)
    arg = j0z[sv - 1] / rD
    soln = lambda a: O.soln(P, D, a, rD, p, zD, zl)
    tmp = []
    for j in range(1, dk.R + 1):
        w, a = O.tanh_sinh(dk.k - dk.R + j, arg)
        f = np.stack([c(soln(ai)) for ai in a])
        tmp.append(arg / 2.0 * np.tensordot(w, f, axes=(0, 0)))
    hv = np.array([4.0 / 2 ** (dk.k - dk.R + j) for j in range(1, dk.R + 1)])
    npp = len(pc)
    tot = np.zeros(npp, complex)
    areas = np.zeros((dk.nacc, npp), complex)
    for jj in range(dk.nacc):
        lob, hib = j0z[sv + jj - 1] / rD, j0z[sv + jj] / rD
        y = ((hib - lob) * gx + (hib + lob)) / 2.0
        f = np.stack([c(soln(yi)) for yi in y])
        areas[jj] = (hib - lob) / 2.0 * np.tensordot(gw, f, axes=(0, 0))[0]
    for i in range(npp):
        e = O.extrap(hv, np.array([[q[0, i].real, q[0, i].imag] for q in tmp]))
        a_, stt = O.wynn(np.stack([areas[:, i].real, areas[:, i].imag], axis=1))
        tot[i] = (e[0] + 1j * e[1]) + (a_[0] + 1j * a_[1])
    inv = lambda v: (O.dehoog(dk.M, dk.alpha, dk.tol, tD, 2 * tD, ri(v)), O.dehoog(dk.M, dk.alpha, dk.tol, tD, 2 * tD, ri(v * pc)) * tD)
    h0, d0 = inv(tot)
    eps = 1e-12
    ah, ad = [], []
    for m in range(npp):
        for dz in (eps, 1j * eps):
            v = tot.copy(); v[m] *= (1.0 + dz)
            h1, d1 = inv(v)
            ah.append(abs(h1 - h0) / abs(h0) / eps); ad.append(abs(d1 - d0) / abs(d0) / eps)
    print(f"row {row} t {t[row]:.4g}: amplification of a relative perturbation of ONE Laplace-space value: h max {max(ah):.3g} (m = {int(np.argmax(ah)) // 2}), "
          f"dh max {max(ad):.3g} (m = {int(np.argmax(ad)) // 2}); sum over the values h {sum(ah) / 2:.3g} dh {sum(ad) / 2:.3g}")
