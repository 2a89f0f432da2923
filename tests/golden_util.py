"""helpers shared by the parity tests: fixture loading and ulp / relative-error measures"""
import json
import os
import struct

import numpy as np

from unconfined_amd.abi import params_from_deck
from unconfined_amd.deck import Deck, TimeSpec

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DECKS = os.path.join(GOLD, "decks")


def deck_names():
    return sorted(f[7:-5] for f in os.listdir(GOLD) if f.startswith("stages_") and f.endswith(".json"))


def unhx(s):
    return struct.unpack("<d", struct.pack("<Q", int(s, 16)))[0]


def load_deck(name):
    dk = Deck.read(os.path.join(DECKS, f"{name}.in"))
    ts = TimeSpec.read(os.path.join(DECKS, dk.timeFileName))
    return dk, ts, params_from_deck(dk)


def load_stages(name):
    with open(os.path.join(GOLD, f"stages_{name}.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLD, f"stages_{name}.npz"))


def load_e2e(name):
    p = os.path.join(GOLD, f"e2e_{name}.npz")
    return np.load(p) if os.path.exists(p) else None


def ulps(a, b):
    """distance in units of last place; NaN==NaN counts as 0, NaN vs number as a huge value"""
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    ai = a.view(np.int64).astype(np.int64)
    bi = b.view(np.int64).astype(np.int64)
    ai = np.where(ai < 0, np.int64(-2 ** 63) - ai, ai)
    bi = np.where(bi < 0, np.int64(-2 ** 63) - bi, bi)
    d = np.abs(ai.astype(np.float64) - bi.astype(np.float64))
    both = np.isnan(a) & np.isnan(b)
    one = np.isnan(a) ^ np.isnan(b)
    d = np.where(both, 0.0, d)
    d = np.where(one, 1e30, d)
    return d


def bits_equal(a, b):
    return float(ulps(a, b).max(initial=0.0)) == 0.0


def crel(a, b, floor=0.0):
    """relative error of complex vectors stored as [...,2] against reference b"""
    a = np.asarray(a); b = np.asarray(b)
    za = a[..., 0] + 1j * a[..., 1]
    zb = b[..., 0] + 1j * b[..., 1]
    with np.errstate(all="ignore"):
        r = np.abs(za - zb) / np.maximum(np.abs(zb), floor if floor > 0 else 1e-300)
    both_bad = ~np.isfinite(za) & ~np.isfinite(zb)
    r = np.where(both_bad, 0.0, r)
    return np.where(np.isnan(r), np.inf, r)


def rel_err(x, ref, floor=1e-3):
    """|x-ref| / max(|ref|, floor)  (measure iii)"""
    x = np.asarray(x, float); ref = np.asarray(ref, float)
    return np.abs(x - ref) / np.maximum(np.abs(ref), floor)


COND_K = 64.0          # Laplace-space values good to COND_K u of the largest: what gate (2) grants per row through c(row)


def conditioning(name, ir):
    """(c_h, c_dh) per row of radius `ir` of deck `name`: first-order amplification of the last stage (de Hoog) of a
    perturbation of the Laplace-space values by epsilon times the largest of them (oracle/gen_conditioning.py)"""
    z = np.load(os.path.join(GOLD, "conditioning.npz"))
    return z[f"ch_{name}_r{ir}"].astype(np.float64), z[f"cdh_{name}_r{ir}"].astype(np.float64)


def e2e_gate_bounds(oracle, name, ir, factor=20.0):
    """per-row bounds of the end-to-end gate (2) of tests/test_gpu_parity.py for radius `ir` of deck `name`:
    max(1e-10, factor x noise, COND_K u c(row)); noise = the larger of the reference's build-to-build spread (running max
    over +-8 rows) and its error against the binary128 evaluation on the truth subsample; c(row) = conditioning of the
    inversion at that time (conditioning()).  Returns (ref rows, bound_h, bound_dh); errors are relative with the floor
    max(|ref|, 1e-3) on the DIMENSIONAL (as printed) values."""
    e2e = load_e2e(name)
    tr = np.load(os.path.join(GOLD, f"truth_{name}.npz"))
    dk, ts, P = load_deck(name)
    D = oracle.nondim(P)
    t = oracle.logspace(ts.min_log, ts.max_log, ts.n)
    tD = t / D.Tc
    sv = oracle.split_vector(list(dk.j0s), tD)
    zz = oracle.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd)
    zD = zz / D.Lc
    zl = oracle.zlay(D, zD)
    rD = np.full_like(tD, float(e2e["radii"][ir]) / D.Lc)
    idx = tr["idx"]
    ho, dho = oracle.batch(P, tD[idx], rD[idx], sv[idx], zD, zl)
    sc = 1.0 if dk.dimless else D.Hc
    fl_raw = 1e-3 / sc
    noise_t = (float(rel_err(ho, tr[f"h_r{ir}"], fl_raw).max()), float(rel_err(dho, tr[f"dh_r{ir}"], fl_raw).max()))
    ref, alt = e2e[f"O2_r{ir}"], e2e[f"O3native_r{ir}"]
    out = []
    cond = conditioning(name, ir)
    for col, nt_, c in ((1, noise_t[0], cond[0]), (2, noise_t[1], cond[1])):
        spread = rel_err(alt[:, col], ref[:, col], 1e-3)
        k = 8
        sp = np.array([spread[max(0, i - k): i + k + 1].max() for i in range(len(spread))])
        out.append(np.maximum(np.maximum(1e-10, factor * np.maximum(sp, nt_)), COND_K * 2.220446049250313e-16 * c))
    return ref, out[0], out[1]
