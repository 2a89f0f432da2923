#!/usr/bin/env python3
"""fast vs faithful flavour on random parameter sets (both on the GPU): a broad net for path-specific slips
(specialised kernels, regime flags, derived primitives).  Prints the worst relative differences."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck
from unconfined_amd import engine
from unconfined_amd.abi import params_from_deck

def run(nsets=40, seed=7, verbose=True, judge_above=1e-6, max_judged=6, models=(1, 3, 4, 5, 6)):
    rng = np.random.default_rng(seed)
    # key -> base deck; the keys >= 10 are further variants of a model (its other evaluator / screen position)
    names = ((0, "c1_theis"), (1, "hantush_lay1"), (2, "hstorage_partpen_lay2"), (3, "c3_moench"), (4, "malama_fullpen"), (5, "neuman74_partpen"),
             (6, "mishra_fd30"), (16, "mishra_malama"), (12, "hstorage_fullpen_lay1"))
    only = os.environ.get("UCF_FUZZ_MODELS")
    models = tuple(int(x) for x in only.split(",")) if only else models
    names = tuple(x for x in names if x[0] in models)
    base = {m: load_deck(n)[0] for m, n in names if os.path.exists(os.path.join(ROOT, "tests", "golden", "decks", n + ".in"))}
    worst = []
    judged = []
    import oracle_lib
    oracle, oracle_q = oracle_lib.Oracle(), oracle_lib.Oracle(quad=True)
    for i in range(nsets):
        model = int(rng.choice(list(base)))
        dk = base[model]
        model = dk.model
        b = dk.b
        full = rng.random() < 0.3
        d = 0.0 if full else b * rng.uniform(0.0, 0.4)
        l = b if full else min(b, d + b * rng.uniform(0.1, 0.6))
        dk2 = dk.replace(Kr=dk.Kr * 10 ** rng.uniform(-1, 1), kappa=10 ** rng.uniform(-1.5, 0.3), Ss=dk.Ss * 10 ** rng.uniform(-1, 1),
                         Sy=min(0.45, dk.Sy * 10 ** rng.uniform(-0.7, 0.3)), l=l, d=d, beta=(0.0 if rng.random() < 0.5 else 10 ** rng.uniform(-2, 1)))
        if dk.model == 2:     # wellbore storage: the well and casing radii and the observation-well delay move too
            rw = dk.rw * 10 ** rng.uniform(-0.5, 0.5)
            dk2 = dk2.replace(rw=rw, rc=rw * 10 ** rng.uniform(-0.3, 0.3), rwobs=dk.rwobs * 10 ** rng.uniform(-0.5, 0.5), sF=dk.sF * 10 ** rng.uniform(-0.5, 0.5))
        if dk.model == 6:     # the unsaturated zone: sorptive numbers, air entry / saturation offsets, thickness
            dk2 = dk2.replace(ac=dk.ac * 10 ** rng.uniform(-0.5, 0.5), ak=dk.ak * 10 ** rng.uniform(-0.5, 0.5), usL=dk.usL * 10 ** rng.uniform(-0.3, 0.3))
        P = params_from_deck(dk2)
        try:
            pf, pg = engine.Plan(P, mode="fast"), engine.Plan(P, mode="faithful")
        except Exception as e:
            if verbose: print("skip", i, e)
            continue
        D = pf.derived
        npts = int(os.environ.get("UCF_FUZZ_NPTS", "48"))
        tD = 10.0 ** rng.uniform(-2, 5, npts); rD = 10.0 ** rng.uniform(-1, 1, npts)
        zD = np.sort(rng.uniform(0.02, 0.98, 2)); zl = pf.zlay(zD)
        sv = pf.split_vector(tD)
        hf, dhf = pf.drawdown(tD, rD, sv, zD, zl)
        hg, dhg = pg.drawdown(tD, rD, sv, zD, zl)
        sc = max(np.nanmax(np.abs(hg)), 1e-300)
        e = np.abs(hf - hg) / np.maximum(np.abs(hg), 1e-4 * sc)
        k = np.unravel_index(np.nanargmax(e), e.shape)
        if float(e[k]) > judge_above and len(judged) < max_judged:
            # who is right?  both flavours and the CPU oracle (= the reference) against the binary128 evaluation
            ho, _ = oracle.batch(P, tD[k[0]:k[0] + 1], rD[k[0]:k[0] + 1], sv[k[0]:k[0] + 1], zD, zl)
            ht, _ = oracle_q.batch(P, tD[k[0]:k[0] + 1], rD[k[0]:k[0] + 1], sv[k[0]:k[0] + 1], zD, zl, threads=8)
            den = max(abs(ht[0, k[1]]), 1e-4 * sc)
            judged.append((i, abs(hf[k] - ht[0, k[1]]) / den, abs(hg[k] - ht[0, k[1]]) / den, abs(ho[0, k[1]] - ht[0, k[1]]) / den))
        ndiff = int(np.sum(np.isnan(hf) != np.isnan(hg)))
        # NaN patterns count as equal when they are, or -- deep in the overflow regime (kappa < 0.05: Re(eta) up to 1500, every
        # value there is the product of the in-band rules acting on overflowed samples) -- when at most 3 % of the values differ
        nan_ok = ndiff == 0 or (float(dk2.kappa) < 0.05 and ndiff <= 0.03 * hf.size)
        if not nan_ok and ndiff <= 0.03 * hf.size:
            # ... or where the reference has lost the answer anyway: at (up to three of) the points in question the CPU oracle
            # is NaN itself or further than 1e-6 from the binary128 evaluation (whether an overflowing intermediate ends as
            # Inf / Inf or as a large finite number is then decided by the last bit)
            lost = []
            for (q, z) in np.argwhere(np.isnan(hf) != np.isnan(hg))[:3]:
                ho, _ = oracle.batch(P, tD[q:q + 1], rD[q:q + 1], sv[q:q + 1], zD, zl)
                ht, _ = oracle_q.batch(P, tD[q:q + 1], rD[q:q + 1], sv[q:q + 1], zD, zl, threads=8)
                lost.append(bool(np.isnan(ho[0, z]) or abs(ho[0, z] - ht[0, z]) > 1e-6 * abs(ht[0, z])))      # (not of the set's scale: that may be garbage itself)
            nan_ok = all(lost)
        worst.append((float(e[k]), i, model, full, float(dk2.kappa), float(rD[k[0]]), float(tD[k[0]]), float(zD[k[1]]), int(zl[k[1]]), bool(nan_ok)))
    worst.sort(reverse=True)
    if verbose:
        for w in worst[:12]:
            print("rel diff %.2e set %d model %d full=%s kappa=%.3g rD=%.3g tD=%.3g zD=%.2f lay=%d nan-pattern-equal=%s" % w)
        es = np.array([w[0] for w in worst])
        for j in judged:
            print("set %d: error vs binary128 truth: fast %.2e  faithful %.2e  reference (CPU oracle) %.2e" % j)
        print("sets", len(es), "median of worst-per-set", np.median(es), "max", es.max(), "nan patterns equal:", all(w[-1] for w in worst))
    return worst, judged

if __name__ == "__main__":
    run(nsets=int(sys.argv[1]) if len(sys.argv) > 1 else 40, seed=int(sys.argv[2]) if len(sys.argv) > 2 else 7)
