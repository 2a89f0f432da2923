#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- mid-pipeline stage vectors (fixture 2): for a few (t, r) points of the
C2 (fully penetrating), C2pp (partially penetrating), C3 (nz = 2), C4 and C5 decks the vectors that the loop body of
driver.f90:129-216 forms between "sample" and "h, dh":

    tmp(R, nz, np)      level sums of the tanh-sinh part, scaled by arg/2                 driver.f90:135,154-156
    finint(nz, np)      their Richardson / Neville extrapolation to h -> 0                 :159-163
    glarea(nacc, nz, np) Gauss-Lobatto areas between successive J0 zeros                   :187-203
    infint(nz, np)      Wynn-epsilon of the areas                                          :205-212
    totlap(nz, np)      finint + infint                                                    :216
    h, dh               the two de Hoog inversions                                         :219-230

from the oracle's ucfo_point (binary64; the oracle is pinned bit-for-bit to the reference, tests/test_oracle_golden.py),
plus the binary128 evaluation of totlap (the truth that arbitrates).  The points lie on a product grid of 128 log-spaced
times x 2 radii per deck so that the same points can be reached through every lane layout of the device path
(tests/test_gpu_stages.py).  Writes tests/golden/midstages.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import GOLD, load_deck  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

DECKS = ["c2_neuman74_fullpen", "neuman74_partpen", "c3_moench", "c4_malama_partpen", "c5_mishra_fd64"]
NT, RADII = 128, (0.3, 2.0)
PICKS = [(4, 0), (37, 1), (70, 0), (101, 1), (126, 0)]        # (time index, radius index)


def grid(O, name):
    dk, ts, P = load_deck(name)
    D = O.nondim(P)
    tD = 10.0 ** O.linspace(-2.0, 4.0, NT)
    sv = O.split_vector(list(dk.j0s), tD)
    zz = O.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd)
    zD = zz / D.Lc
    return dk, P, D, tD, sv, np.array(RADII), zD, O.zlay(D, zD)


def main():
    O, Q = Oracle(), Oracle(quad=True)
    out = {"picks": np.array(PICKS, np.int32), "radii": np.array(RADII), "nt": np.array([NT])}
    for name in DECKS:
        dk, P, D, tD, sv, rD, zD, zl = grid(O, name)
        j0z = O.j0_zeros(D.nj0z)
        for k, (it, ir) in enumerate(PICKS):
            h, dh, st = O.point(P, D, j0z, tD[it], rD[ir], sv[it], zD, zl, stages=True)
            hq, dhq, sq = Q.point(P, D, j0z, tD[it], rD[ir], sv[it], zD, zl, stages=True)
            for key in ("tmp", "finint", "glarea", "infint", "totlap"):
                out[f"{name}_{k}_{key}"] = st[key]
            out[f"{name}_{k}_totlap_q"] = sq["totlap"]
            out[f"{name}_{k}_tmp_q"] = sq["tmp"]
            out[f"{name}_{k}_glarea_q"] = sq["glarea"]
            out[f"{name}_{k}_h"], out[f"{name}_{k}_dh"] = h, dh
            print(name, k, h, flush=True)
    np.savez_compressed(os.path.join(GOLD, "midstages.npz"), **out)


if __name__ == "__main__":
    main()
