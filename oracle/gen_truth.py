#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- binary128 "truth" for the end-to-end parity gate.

The binary64 oracle is pinned bit-for-bit to the reference (tests/test_oracle_golden.py).
The same source compiled in binary128 (libucf_oracle_q.so) evaluates the *same algorithm*
(same branches, MAXEXP of the binary64 build, same in-band rules) essentially without
rounding error.  It arbitrates where the reference's own rounding noise (amplified ~1e5-1e7x
by Wynn-epsilon + de Hoog) exceeds 1e-10: a device result is as good as the reference's if it
is as close to this truth as the reference is.

Writes tests/golden/truth_<deck>.npz: for a strided subsample of the deck's times and every
radius of the e2e fixture: idx, raw dimensionless h/dh [n, nz] in binary128 rounded to double.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import GOLD, deck_names, load_deck, load_e2e  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

NSUB = {"c2_neuman74_fullpen": 32, "c3_moench": 24, "c4_malama_partpen": 24, "c5_mishra_fd64": 12,
        "malama_k10": 8, "mishra_malama": 8, "mishra_fd30": 16}


def main():
    Q, O = Oracle(quad=True), Oracle()
    names = sys.argv[1:] or deck_names()
    for name in names:
        dk, ts, P = load_deck(name)
        e2e = load_e2e(name)
        D = O.nondim(P)
        t = O.logspace(ts.min_log, ts.max_log, ts.n)
        tD = t / D.Tc
        sv = O.split_vector(list(dk.j0s), tD)
        zz = O.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd)
        zD = zz / D.Lc
        zl = O.zlay(D, zD)
        nsub = NSUB.get(name, 24)
        idx = np.unique(np.linspace(0, len(t) - 1, nsub).astype(int))
        arrs = {"idx": idx}
        for ir, r in enumerate(e2e["radii"]):
            rD = np.full(len(idx), float(r) / D.Lc)
            h, dh = Q.batch(P, tD[idx], rD, sv[idx], zD, zl, threads=8)
            arrs[f"h_r{ir}"] = h
            arrs[f"dh_r{ir}"] = dh
            print(name, ir, "done", flush=True)
        np.savez_compressed(os.path.join(GOLD, f"truth_{name}.npz"), **arrs)


if __name__ == "__main__":
    main()
