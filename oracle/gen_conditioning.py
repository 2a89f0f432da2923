#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- conditioning of the last stage, per row of every end-to-end fixture.

De Hoog's inversion (invlap.f90:46-141) is a Pade-type accelerator: at single times it amplifies a relative perturbation
of the 2M+1 Laplace-space values totlap_m by 1e3 ... 1e6 (the reference's own -O2 / -O3 outputs show such rows).  For
every row (time) of every radius of tests/golden/e2e_<deck>.npz this script evaluates, with the oracle,

    c_h(row)  = sum_m |d h / d totlap_m| max_k |totlap_k| / max(|h|, floor)          (and c_dh for t dh/dt)

the first-order bound on the relative change of the result when every Laplace-space value moves by epsilon times the
LARGEST of them.  That is the noise model that fits what is observed: the values come out of sums
over the abscissae and out of the series acceleration with an error floor set by the largest terms, and the small values
at high Laplace index are where the inversion is most sensitive -- the row that broke the round-2 gate (C2, radius 2, row
168: dh off by 2.4e-10 with Laplace-space values 3e-15 of the largest away from the oracle's) has c_dh = 4.1e4, its
neighbours the same; with relative perturbations of each value's own modulus the same row has c = 5e3, which would need
values 50 u off.  (|d/d totlap_m| = norm of the derivatives along the real and the imaginary direction, central differences
of the oracle's de Hoog; floor = 1e-3 in the printed units, as in the parity gates.)  A device result whose Laplace-space
values are good to k u max|totlap| cannot be expected closer than k u c(row): gate (2) of tests/test_gpu_parity.py uses
max(1e-10, 20 x reference noise, k u c(row)) per row -- no exception clause.
Writes tests/golden/conditioning.npz (ch_<deck>_r<ir>, cdh_<deck>_r<ir>: [rows])."""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import GOLD, deck_names, load_deck, load_e2e  # noqa: E402


def one(args):
    name, ir = args
    from oracle_lib import Oracle
    O = Oracle()
    dk, ts, P = load_deck(name)
    e2e = load_e2e(name)
    D = O.nondim(P)
    t = O.logspace(ts.min_log, ts.max_log, ts.n)
    tD = t / D.Tc
    sv = O.split_vector(list(dk.j0s), tD)
    zz = O.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd)
    zD = zz / D.Lc
    zl = O.zlay(D, zD)
    rD = float(e2e["radii"][ir]) / D.Lc
    j0z = O.j0_zeros(D.nj0z)
    sc = 1.0 if dk.dimless else D.Hc
    floor = 1e-3 / sc
    nzq = 1 if dk.piezometer else dk.zOrd
    ch, cd = np.zeros(len(t)), np.zeros(len(t))
    for row in range(len(t)):
        h, dh, st = O.point(P, D, j0z, tD[row], rD, sv[row], zD, zl, stages=True)
        p = O.pvalues(2 * tD[row], dk.M, dk.alpha, dk.tol)
        pc = p[:, 0] + 1j * p[:, 1]
        worst_h = worst_d = 0.0
        for z in range(nzq):                       # (a screened well averages its depths: the worst depth stands for the row)
            tl = st["totlap"][z, :, 0] + 1j * st["totlap"][z, :, 1]
            if not np.isfinite(tl).all():
                worst_h = worst_d = np.inf
                continue

            def inv(v):
                a = O.dehoog(dk.M, dk.alpha, dk.tol, tD[row], 2 * tD[row], np.stack([v.real, v.imag], axis=1))
                w = v * pc
                b = O.dehoog(dk.M, dk.alpha, dk.tol, tD[row], 2 * tD[row], np.stack([w.real, w.imag], axis=1)) * tD[row]
                return a, b
            sh = sd = 0.0
            eps = 1e-7
            big = np.abs(tl).max()
            if big == 0:
                continue
            for m in range(len(tl)):
                g = []
                for dz in (eps * big, 1j * eps * big):
                    vp = tl.copy(); vp[m] += dz
                    vm = tl.copy(); vm[m] -= dz
                    (hp, dp_), (hm, dm) = inv(vp), inv(vm)
                    g.append(((hp - hm) / (2 * eps), (dp_ - dm) / (2 * eps)))
                sh += np.hypot(g[0][0], g[1][0])
                sd += np.hypot(g[0][1], g[1][1])
            worst_h = max(worst_h, sh / max(abs(h[z]), floor))
            worst_d = max(worst_d, sd / max(abs(dh[z]), floor))
        ch[row], cd[row] = worst_h, worst_d
    return name, ir, ch, cd


def main():
    names = sys.argv[1:] or deck_names()
    jobs = [(n, ir) for n in names for ir in range(len(load_e2e(n)["radii"]))]
    out = {}
    path = os.path.join(GOLD, "conditioning.npz")
    if os.path.exists(path) and sys.argv[1:]:
        out = dict(np.load(path))
    with ProcessPoolExecutor(max_workers=8) as ex:
        for name, ir, ch, cd in ex.map(one, jobs):
            out[f"ch_{name}_r{ir}"] = ch.astype(np.float32)
            out[f"cdh_{name}_r{ir}"] = cd.astype(np.float32)
            print(name, ir, "c_h median %.3g max %.3g   c_dh median %.3g max %.3g" % (np.median(ch), ch.max(), np.median(cd), cd.max()), flush=True)
    np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
