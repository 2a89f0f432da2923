"""fast / faithful flavour against the binary128 evaluation and the binary64 oracle on an end-to-end fixture:
per radius the worst point of gate (1) of tests/test_gpu_parity.py.  usage: dbg_truth.py <deck name> [mode]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from golden_util import load_e2e, rel_err
from unconfined_amd import engine
import oracle_lib
oracle = oracle_lib.Oracle()
name = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "fast"
e2e, tr = load_e2e(name), T._truth(name)
idx = tr["idx"]
for ir in range(len(e2e["radii"])):
    dk, P, D, t, tD, rD, sv, zD, zl = T._grid(oracle, name, ir, e2e)
    plan = engine.Plan(P, mode=mode)
    h, dh = plan.drawdown(tD[idx], rD[idx], sv[idx], zD, zl)
    ho, dho = oracle.batch(P, tD[idx], rD[idx], sv[idx], zD, zl)
    floor = 1e-3 / (1.0 if dk.dimless else D.Hc)
    for got, ref, truth, label in ((h, ho, tr[f"h_r{ir}"], "h"), (dh, dho, tr[f"dh_r{ir}"], "dh")):
        eg, er = rel_err(got, truth, floor), rel_err(ref, truth, floor)
        i = np.unravel_index(int(np.argmax(eg)), eg.shape)
        print(f"{name} {mode} r{ir} {label}: worst {eg.max():.3e} at pt {i} tD {tD[idx][i[0]]:.4g} (ref there {er[i]:.3e}); ref worst {er.max():.3e}; "
              f"median {np.median(eg):.2e} / ref {np.median(er):.2e}; ratio to bound {eg.max() / max(1e-10, 20 * er.max()):.2f}")

# gate (2): every row of the reference binary's .out, bound from the local O2/O3 spread
from unconfined_amd.host import screen_average_np
for ir in range(len(e2e["radii"])):
    dk, P, D, t, tD, rD, sv, zD, zl = T._grid(oracle, name, ir, e2e)
    plan = engine.Plan(P, mode=mode)
    h, dh = plan.drawdown(tD, rD, sv, zD, zl)
    sc = 1.0 if dk.dimless else D.Hc
    hobs, dobs = screen_average_np(h, dk) * sc, screen_average_np(dh, dk) * sc
    ref, alt = e2e[f"O2_r{ir}"], e2e[f"O3native_r{ir}"]
    ho, dho = oracle.batch(P, tD[idx], rD[idx], sv[idx], zD, zl)
    fl_raw = 1e-3 / sc
    noise_t = {"h": float(rel_err(ho, tr[f"h_r{ir}"], fl_raw).max()), "dh": float(rel_err(dho, tr[f"dh_r{ir}"], fl_raw).max())}
    for col, got, label in ((1, hobs, "h"), (2, dobs, "dh")):
        err = rel_err(got, ref[:, col], 1e-3)
        spread = rel_err(alt[:, col], ref[:, col], 1e-3)
        sp = np.array([spread[max(0, i - 8): i + 9].max() for i in range(len(spread))])
        bound = np.maximum(1e-10, 20.0 * np.maximum(sp, noise_t[label]))
        ratio = err / bound
        w = np.argsort(ratio)[::-1][:3]
        print(f"{name} {mode} r{ir} {label} vs .out: noise_t {noise_t[label]:.2e}; worst rows " +
              "; ".join(f"row {i} t {t[i]:.4g} err {err[i]:.2e} bound {bound[i]:.2e} ratio {ratio[i]:.2f} spread@row {spread[i]:.1e}" for i in w))
