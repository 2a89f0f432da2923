"""exploratory GPU run: prints parity statistics of every stage against the oracle"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import deck_names, load_deck, load_stages, load_e2e, ulps, crel, GOLD
from oracle_lib import Oracle
from unconfined_amd import engine
from unconfined_amd.engine import Plan

O = Oracle()
print("fp64 FMA peak: %.2f TFLOP/s" % engine.fp64_fma_peak(), flush=True)
z = np.load(os.path.join(GOLD, "stages_generic.npz"))
nw, ne, nd = z["counts"]
# wynn
by_n = {}
for i in range(nw):
    s = z[f"wynn_in_{i}"]; by_n.setdefault(len(s), []).append(i)
bad = 0
for n, idx in by_n.items():
    ser = np.stack([z[f"wynn_in_{i}"] for i in idx])
    acc, st = engine.wynn_epsilon(ser)
    for k, i in enumerate(idx):
        u = ulps(acc[k], z[f"wynn_out_{i}"]).max()
        if u: bad += 1; print("wynn", i, n, acc[k], z[f"wynn_out_{i}"], st[k], u)
print("wynn mismatches:", bad, "of", nw, flush=True)
bad = 0
for i in range(ne):
    out = engine.extraptozero(z[f"extrap_x_{i}"], z[f"extrap_y_{i}"][None])
    u = ulps(out[0], z[f"extrap_out_{i}"]).max()
    if u: bad += 1; print("extrap", i, out[0], z[f"extrap_out_{i}"], u)
print("extrap mismatches:", bad, "of", ne, flush=True)
mxu = 0; mxr = 0
for i in range(nd):
    M, alpha, tol, t, tee = z[f"dehoog_par_{i}"]
    out = engine.dehoog(int(M), alpha, tol, t, tee, z[f"dehoog_fp_{i}"])
    ref = z[f"dehoog_out_{i}"][0]
    u = ulps(out, np.array([ref])).max()
    r = abs(out[0] - ref) / max(abs(ref), 1e-300) if np.isfinite(ref) else (0 if np.isnan(out[0]) == np.isnan(ref) else 9)
    mxu = max(mxu, u); mxr = max(mxr, r)
    if r > 1e-10: print("dehoog", i, M, t, out[0], ref, r)
print("dehoog max ulp", mxu, "max rel", mxr, flush=True)

names = sys.argv[1:] or deck_names()
for name in names:
    dk, ts, P = load_deck(name)
    meta, zz = load_stages(name)
    for mode in ("faithful", "fast"):
        try:
            plan = Plan(P, mode=mode)
        except Exception as e:
            print(name, "plan failed:", e); break
        D = plan.derived
        if mode == "faithful":
            Do = O.nondim(P)
            same = all(getattr(D, f) == getattr(Do, f) for f, _ in type(D)._fields_ if f != "MoenchGamma")
            j0ok = ulps(plan.j0z(), O.j0_zeros(D.nj0z)).max()
            w, x = plan.tanh_sinh(P.R); wo, ao = O.tanh_sinh(P.k, 2.0)
            glx, glw = plan.gauss_lobatto(); gx, gw = O.gauss_lobatto(P.ord)
            print(f"{name}: derived same={same} j0z ulp={j0ok} tsw ulp={ulps(w,wo).max()} tsx*2/2 ulp={ulps(x*2.0/2.0,ao).max()} gl ulp={max(ulps(glx,gx).max(),ulps(glw,gw).max())}")
        zD = zz["par_zD"]; zl = zz["par_zLay"]
        worst = 0; worstu = 0; nanmis = 0
        t0 = time.time()
        for i, (tD, a, rD) in enumerate(zip(zz["soln_tD"], zz["soln_a"], zz["soln_rD"])):
            p = O.pvalues(2 * tD, dk.M, dk.alpha, dk.tol)
            try:
                fp = plan.lap_hank_soln([a], rD, p, zD, zl)[0]
            except Exception as e:
                print("   sample failed:", e); break
            ref = zz["soln_fp"][i]
            nanmis += int((np.isnan(fp) != np.isnan(ref)).sum())
            r = crel(fp, ref)
            r = np.where(np.isfinite(r), r, 0)
            if r.max() > worst: worst = r.max(); wi = (a, tD)
            worstu = max(worstu, np.where(np.isnan(ref), 0, ulps(fp, ref)).max())
        print(f"   {mode}: samples max rel {worst:.2e} at {wi} max ulp {worstu:.0f} nan-mismatch {nanmis}", flush=True)
        # end to end on a subsample
        e2e = load_e2e(name)
        if e2e is None: continue
        t = O.logspace(ts.min_log, ts.max_log, ts.n); tD = t / D.Tc
        sv = O.split_vector(list(dk.j0s), tD)
        NSUB = {"c2_neuman74_fullpen": 32, "c3_moench": 24, "c4_malama_partpen": 24, "c5_mishra_fd64": 12,
                "malama_k10": 8, "mishra_malama": 8, "mishra_fd30": 16}
        idx = np.unique(np.linspace(0, len(t) - 1, NSUB.get(name, 24)).astype(int))
        for ir in range(len(e2e["radii"])):
            rD = np.full(len(idx), e2e["radii"][ir] / D.Lc)
            t0 = time.time()
            try:
                h, dh, st = plan.drawdown(tD[idx], rD, sv[idx], zD, zl, with_stats=True)
            except Exception as e:
                print("   drawdown failed:", e); break
            dt = time.time() - t0
            ho, dho = O.batch(P, tD[idx], rD, sv[idx], zD, zl)
            fl = 1e-3 / (1 if dk.dimless else D.Hc)
            eh = np.abs(h - ho) / np.maximum(np.abs(ho), fl)
            ed = np.abs(dh - dho) / np.maximum(np.abs(dho), fl)
            msg = ""
            tp = os.path.join(GOLD, f"truth_{name}.npz")
            if os.path.exists(tp):
                tr = np.load(tp)
                if np.array_equal(tr["idx"], idx):
                    ht, dht = tr[f"h_r{ir}"], tr[f"dh_r{ir}"]
                    f = lambda x, y: (np.abs(x - y) / np.maximum(np.abs(y), fl))
                    msg = (f" | vs truth: gpu h {f(h,ht).max():.1e}/{np.median(f(h,ht)):.1e} ref h {f(ho,ht).max():.1e}/{np.median(f(ho,ht)):.1e}"
                           f" gpu dh {f(dh,dht).max():.1e}/{np.median(f(dh,dht)):.1e} ref dh {f(dho,dht).max():.1e}/{np.median(f(dho,dht)):.1e}")
            print(f"   {mode} r={e2e['radii'][ir]:8.3f}: vs oracle h {eh.max():.2e} dh {ed.max():.2e} ({dt*1e3:.0f} ms){msg}", flush=True)
