#!/usr/bin/env python3
"""fast vs faithful flavour on random SHAPES: numerical settings (de Hoog M, tanh-sinh k / R, accelerated zeros, Gauss-Lobatto
order, J0 split range), depths (1 ... 4, any layers) and call shapes (lists of 1 ... 700 points, grids of 1 ... 200 times x
1 ... 9 radii) -- the index arithmetic of every lane layout, not the formulas.  Meant to run under UCF_GUARD=1.
usage: fuzz_shapes.py NSETS SEED"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck
from unconfined_amd import engine
from unconfined_amd.abi import params_from_deck

DECKS = ("c1_theis", "hantush_lay1", "hantush_screen", "hstorage_partpen_lay2", "c3_moench", "malama_fullpen", "neuman74_partpen",
         "c2_neuman74_fullpen", "mishra_fd30", "mishra_malama")

def run(nsets=60, seed=3, verbose=True, max_judged=8):
    rng = np.random.default_rng(seed)
    worst = []; judged = []
    for i in range(nsets):
        name = DECKS[int(rng.integers(len(DECKS)))]
        dk = load_deck(name)[0]
        k = int(rng.integers(3, 9)); R = int(rng.integers(1, k))
        lo = int(rng.integers(1, 4)); hi = lo + int(rng.integers(0, 3))
        dk2 = dk.replace(M=int(rng.choice([3, 8, 15, 20, 31, 32, 40, 70])), k=k, R=R, j0s=[lo, hi],
                         nacc=int(rng.choice([2, 5, 10, 12, 13, 16])), ord=int(rng.choice([4, 7, 12, 25, 50, 81])))
        try:
            P = params_from_deck(dk2)
            pf, pg = engine.Plan(P, mode="fast"), engine.Plan(P, mode="faithful")
        except Exception as e:
            if verbose: print("skip", i, name, str(e)[:80])
            continue
        nz = int(rng.integers(1, 5))
        zD = np.sort(rng.uniform(0.02, 0.98, nz)); zl = pf.zlay(zD)
        kind = "list" if rng.random() < 0.5 else "grid"
        if kind == "list":
            n = int(rng.choice([1, 5, 63, 64, 65, 255, 256, 300, 700]))
            tD = 10.0 ** rng.uniform(-1, 4, n); rD = 10.0 ** rng.uniform(-0.7, 0.7, n)
            sv = pf.split_vector(tD)
            hf, dhf = pf.drawdown(tD, rD, sv, zD, zl); hg, dhg = pg.drawdown(tD, rD, sv, zD, zl)
            shape = "list %d" % n
        else:
            nt = int(rng.choice([1, 3, 63, 64, 65, 130, 200])); nr = int(rng.integers(1, 10))
            tD = np.sort(10.0 ** rng.uniform(-1, 4, nt)); rD = np.sort(10.0 ** rng.uniform(-0.7, 0.7, nr))
            sv = pf.split_vector(tD)
            hf, dhf = pf.drawdown_grid(tD, sv, rD, zD, zl); hg, dhg = pg.drawdown_grid(tD, sv, rD, zD, zl)
            shape = "grid %dx%d" % (nt, nr)
        sc = max(np.nanmax(np.abs(hg)) if np.isfinite(hg).any() else 0.0, 1e-300)
        e = np.abs(hf - hg) / np.maximum(np.abs(hg), 1e-4 * sc)
        if name.startswith(("hantush", "hstorage")):
            # above the screen the reference's Hantush factor cancels: the faithful flavour reproduces
            # that noise, the fast one does not -- those depths say nothing about the index arithmetic
            e = e[..., np.asarray(zl) != 3]
        emax = float(np.nanmax(e)) if e.size and np.isfinite(e).any() else 0.0
        ndiff = int(np.sum(np.isnan(hf) != np.isnan(hg)))
        if emax > 1e-6 and len(judged) < max_judged:
            # both flavours and the CPU oracle (= the reference) against the binary128 evaluation at the worst point
            import oracle_lib
            full = np.abs(hf - hg) / np.maximum(np.abs(hg), 1e-4 * sc)
            if name.startswith(("hantush", "hstorage")): full[..., np.asarray(zl) == 3] = 0.0
            w = np.unravel_index(np.nanargmax(full), full.shape)
            it, ir = (w[0], w[0]) if kind == "list" else (w[0], w[1])
            t1, r1, s1 = tD[it:it + 1], rD[ir:ir + 1], sv[it:it + 1]
            ho, _ = oracle_lib.Oracle().batch(P, t1, r1, s1, zD, zl)
            ht, _ = oracle_lib.Oracle(quad=True).batch(P, t1, r1, s1, zD, zl, threads=8)
            den = max(abs(ht[0, w[-1]]), 1e-4 * sc)
            judged.append((i, abs(hf[w] - ht[0, w[-1]]) / den, abs(hg[w] - ht[0, w[-1]]) / den, abs(ho[0, w[-1]] - ht[0, w[-1]]) / den))
        worst.append((emax, i, name, shape, "M=%d k=%d R=%d j0s=%d..%d nacc=%d ord=%d nz=%d lay=%s" % (dk2.M, k, R, lo, hi, dk2.nacc, dk2.ord, nz, list(zl)), ndiff, hf.size))
    worst.sort(reverse=True)
    if verbose:
        for w in worst[:10]: print("rel diff %.2e set %d %s %s %s nan-diff %d of %d" % w)
        for j in judged: print("set %d: error vs binary128 truth: fast %.2e  faithful %.2e  reference (CPU oracle) %.2e" % j)
        print("sets", len(worst), "median", float(np.median([w[0] for w in worst])), "max", worst[0][0], "nan-diff sets", sum(1 for w in worst if w[5]))
    return worst, judged

def run_multi(nsets=30, seed=3, verbose=True, judge_above=1e-9):
    """parameter batches (ucf_drawdown_multi): 2 ... 9 plans x 1 ... 300 shared observation points x 1 ... 3 depths with random
    numerical settings against every plan's own call.  The shared launch sequence runs other instantiations of the kernels
    than a single plan's call (parameter blocks in memory, often another lane layout): same formulas, but the compiler is
    free to contract a product and a sum into an FMA in one and not in the other -- the results agree to the rounding of the
    fast flavour (a few 1e-12 of the solution's scale after the accelerations), not always bit for bit.
    Returns (sets run, sets not bit-equal, largest difference relative to the plan's largest |h|)."""
    rng = np.random.default_rng(seed)
    bad = 0; done = 0; emax = 0.0; judged = []
    for i in range(nsets):
        name = ("hantush_lay1", "c3_moench", "malama_fullpen", "neuman74_partpen", "mishra_fd30")[int(rng.integers(5))]
        dk = load_deck(name)[0]
        k = int(rng.integers(4, 8)); R = int(rng.integers(1, k - 1))
        dk = dk.replace(M=int(rng.choice([5, 15, 20, 31])), k=k, R=R, nacc=int(rng.choice([5, 10, 12])), ord=int(rng.choice([7, 25, 50])))
        npl = int(rng.integers(2, 10))
        try:
            plans = [engine.Plan(params_from_deck(dk.replace(Kr=dk.Kr * 10 ** rng.uniform(-0.3, 0.3), kappa=dk.kappa * 10 ** rng.uniform(-0.3, 0.3))), mode="fast") for _ in range(npl)]
        except Exception as e:
            if verbose: print("skip", i, name, str(e)[:80])
            continue
        n = int(rng.choice([1, 7, 31, 32, 33, 64, 65, 130, 300])); nz = int(rng.integers(1, 4))
        D0 = plans[0].derived
        t = D0.Tc * 10.0 ** rng.uniform(-1, 4, n); r = D0.Lc * 10.0 ** rng.uniform(-0.7, 0.7, n); z = np.sort(rng.uniform(0.05, 0.95, nz)) * dk.b
        hm, dhm = engine.drawdown_multi(plans, t, r, z)
        same = True; e_set = 0.0
        for q, pl in enumerate(plans):
            D = pl.derived
            tD, rD, zD = t / D.Tc, r / D.Lc, z / D.Lc
            h1, dh1 = pl.drawdown(tD, rD, pl.split_vector(tD), zD, pl.zlay(zD))
            h1 = h1 * D.Hc; dh1 = dh1 * D.Hc
            same &= np.array_equal(hm[q], h1, equal_nan=True) and np.array_equal(dhm[q], dh1, equal_nan=True)
            if not np.array_equal(np.isnan(hm[q]), np.isnan(h1)): e_set = np.inf
            for which, (a, b) in enumerate(((hm[q], h1), (dhm[q], dh1))):
                sc = np.nanmax(np.abs(b)) if np.isfinite(b).any() else 1.0
                d = np.abs(a - b)
                if np.isfinite(d).any():
                    e_here = float(np.nanmax(d)) / max(sc, 1e-300)
                    if e_here > max(e_set, judge_above):
                        # who is nearer the exact value?  (binary128 evaluation of the same algorithm, tests/oracle_lib.py)
                        import oracle_lib
                        w = np.unravel_index(np.nanargmax(d), d.shape)
                        zl = pl.zlay(zD)
                        ht, dht = oracle_lib.Oracle(quad=True).batch(pl.params, tD[w[0]:w[0] + 1], rD[w[0]:w[0] + 1], pl.split_vector(tD[w[0]:w[0] + 1]), zD, zl, threads=8)
                        tr = (dht if which else ht)[0, w[1]] * D.Hc
                        judged.append((i, q, "dh" if which else "h", e_here, abs(a[w] - tr) / sc, abs(b[w] - tr) / sc))
                    e_set = max(e_set, e_here)
        done += 1; bad += (not same); emax = max(emax, e_set)
        if verbose and not same:
            print("not bit-equal: set", i, name, "plans", npl, "points", n, "nz", nz, "M", dk.M, "k", k, "R", R, "nacc", dk.nacc, "ord", dk.ord, "max diff / scale %.2e" % e_set)
    if verbose:
        for j in judged: print("set %d plan %d %s: batch vs single %.2e of the scale; error vs binary128: batch %.2e  single %.2e" % j)
        print("multi sets", done, "not bit-equal", bad, "largest difference / scale %.2e" % emax)
    return done, bad, emax, judged

if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "multi":
        done, bad, emax, judged = run_multi(nsets=int(sys.argv[1]), seed=int(sys.argv[2])); sys.exit(0 if emax < 1e-9 else 1)

    run(nsets=int(sys.argv[1]) if len(sys.argv) > 1 else 60, seed=int(sys.argv[2]) if len(sys.argv) > 2 else 3)
