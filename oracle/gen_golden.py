#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generates the golden fixtures under tests/golden/.

Runs ONLY in the build container (it needs the reference built by
`make -C oracle ref`, i.e. /root/reference + flang).  It
  1. writes the fixture decks (tests/golden/decks/*.in + time files): plain data,
     the parameter sets of SURVEY.md section 8d (C1..C5) and of the reference's
     runnable example decks, re-typed in our own deck writer;
  2. feeds each deck to oracle/_ref/O2/ref_harness (the reference's own
     read_input, lap_hank_soln, deHoog_*, tanh_sinh_setup, gauss_lobatto_setup,
     wynn_epsilon, extraptozero) and stores inputs + outputs bit-exactly
     (tests/golden/stages_<deck>.npz + .json);
  3. runs the reference binary (flavours O2 and O3native) on each deck, one run
     per radius (time-series mode) and stores the parsed
     .out columns (tests/golden/e2e_<deck>.npz).
The committed fixtures are what tests/ and the GPU box use; nothing at test time
reads /root/reference.
"""
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from unconfined_amd.deck import Deck, TimeSpec  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DECKS = os.path.join(GOLD, "decks")
REFBIN = {f: os.path.join(HERE, "_ref", f) for f in ("O2", "O3native")}


def hx(x: float) -> str:
    return "%016X" % struct.unpack("<Q", struct.pack("<d", float(x)))[0]


def unhx(s: str) -> float:
    return struct.unpack("<d", struct.pack("<Q", int(s, 16)))[0]


# --------------------------------------------------------------------------- decks
CAPE = dict(quiet=0, model=5, dimless=False, timeseries=True, piezometer=True,
            Q=42.8, l=60.2, d=13.2, rw=0.3333, rc=0.3333, gammaSkin=1.0, timeType=1, timePar=[0.0, 1.0],
            b=160.0, Kr=0.24, kappa=0.4, Ss=2.9e-5, Sy=0.23, beta=0.0, MoenchM=1, MoenchAlpha=[-9999.0],
            ac=2.9, ak=0.37, psia=2.0, psik=0.22, usL=20.0, MNtype=2, order=5,
            M=26, alpha=1.0e-8, tol=1.0e-9, k=6, R=4, j0s=[1, 1], nacc=10, ord=50,
            tval=9999.9, rval=85.1, zTop=146.7, zBot=144.7, zOrd=2, rwobs=0.167, sF=1.0)
SMALL = dict(quiet=0, model=1, dimless=True, timeseries=True, piezometer=True,
             Q=2.0e-2, l=5.5, d=4.5, rw=2.54e-2, rc=2.54e-2, gammaSkin=1.0, timeType=1, timePar=[0.0, 1.0],
             b=10.0, Kr=1.0e-4, kappa=0.1, Ss=1.0e-6, Sy=0.25, beta=10.0, MoenchM=1, MoenchAlpha=[-9999.0],
             ac=2.95, ak=0.37, psia=2.0, psik=0.22, usL=20.0, MNtype=2, order=5,
             M=10, alpha=1.0e-8, tol=1.0e-9, k=7, R=5, j0s=[1, 1], nacc=10, ord=50,
             tval=15.5, rval=5.0, zTop=10.0, zBot=0.0, zOrd=5, rwobs=2.54e-2, sF=20.0)
MALAMA = dict(quiet=0, model=5, dimless=True, timeseries=True, piezometer=True,
              Q=2.0189e-2, l=50.288, d=0.02332, rw=2.54e-2, rc=2.54e-2, gammaSkin=1.0, timeType=1, timePar=[0.0, 1.0],
              b=52.669, Kr=1.225e-3, kappa=0.5288, Ss=3.766e-6, Sy=0.2521, beta=2.0, MoenchM=0, MoenchAlpha=[],
              ac=2.95, ak=0.37, psia=2.0, psik=0.22, usL=20.0, MNtype=2, order=5,
              M=26, alpha=1.0e-8, tol=1.0e-9, k=7, R=5, j0s=[2, 2], nacc=12, ord=50,
              tval=15.5, rval=6.5837, zTop=26.33, zBot=0.0, zOrd=5, rwobs=2.54e-2, sF=20.0)


def mk(base, **kw):
    d = dict(base)
    d.update(kw)
    return d


# name -> (deck fields, TimeSpec, radii for the end-to-end runs [None = deck's rval])
DECKSET = {
    # BASELINE.json configs
    "c1_theis": (mk(SMALL, model=0, l=7.5, d=2.5, zTop=1.0), TimeSpec(True, -1, 6, 64), [None]),
    "c2_neuman74_fullpen": (mk(CAPE, l=160.0, d=0.0), TimeSpec(True, -1, 8, 1024),
                            [16.0, 29.6, 54.9, 84.8, 160.0, 480.0, 960.0, 1600.0]),
    "c3_moench": (mk(CAPE, model=3, piezometer=False, b=168.9, Kr=0.2331, kappa=0.6083, Ss=1.305e-5, Sy=0.266,
                     beta=1.0, MoenchM=3, MoenchAlpha=[2.78e-4, 1.68e-2, 4.16e-1]),
                  TimeSpec(True, -1, 8, 256), [16.89, 85.1, 506.7, 1689.0]),
    "c4_malama_partpen": (MALAMA, TimeSpec(True, -1, 8, 256), [5.2669, 6.5837, 52.669, 526.69]),
    "c5_mishra_fd64": (mk(CAPE, model=6, beta=1.0, ac=0.8, ak=0.7, psia=2.0, psik=1.25, usL=19.8, MNtype=2, order=64),
                       TimeSpec(True, -1, 8, 128), [16.0, 85.1, 480.0, 1600.0]),
    # the reference's runnable example parameter sets (100 times each)
    "neuman74_partpen": (CAPE, TimeSpec(True, -1, 8, 100), [None]),
    "malama_partpen_b1": (mk(CAPE, beta=1.0), TimeSpec(True, -1, 8, 100), [None]),
    "mishra_fd30": (mk(CAPE, model=6, beta=1.0, ac=0.8, ak=0.7, psia=2.0, psik=1.25, usL=19.8, MNtype=2, order=30),
                    TimeSpec(True, -1, 8, 100), [None]),
    "mishra_malama": (dict(quiet=0, model=6, dimless=True, timeseries=True, piezometer=True,
                           Q=6.309e-4, l=20.0, d=0.0, rw=1.0e-7, rc=1.0e-7, gammaSkin=1.0, timeType=1, timePar=[0.0, 1.0],
                           b=20.0, Kr=1.0e-4, kappa=1.0, Ss=1.0e-4, Sy=0.30, beta=0.0, MoenchM=0, MoenchAlpha=[],
                           ac=0.5, ak=0.5, psia=2.5e-2, psik=2.0e-2, usL=10.0, MNtype=1, order=20,
                           M=15, alpha=1.0e-7, tol=1.0e-8, k=9, R=7, j0s=[1, 1], nacc=10, ord=100,
                           tval=15.5, rval=5.0, zTop=20.0, zBot=0.0, zOrd=5, rwobs=1.0e-7, sF=1.0),
                      TimeSpec(True, -1, 8, 50), [None]),
    "hantush_fullpen": (mk(SMALL, l=1.0, d=0.0, zTop=1.0), TimeSpec(True, -1, 8, 100), [None]),
    "hantush_lay1": (mk(SMALL, zTop=4.0, zBot=0.0), TimeSpec(True, -1, 8, 50), [None]),     # z=2  below screen
    "hantush_lay2": (mk(SMALL, zTop=10.0, zBot=0.0), TimeSpec(True, -1, 8, 50), [None]),    # z=5  beside screen
    "hantush_lay3": (mk(SMALL, zTop=10.0, zBot=6.0), TimeSpec(True, -1, 8, 50), [None]),    # z=8  above screen
    "hantush_screen": (mk(SMALL, piezometer=False, zTop=9.0, zBot=1.0, zOrd=5), TimeSpec(True, -1, 8, 30), [None]),
    "malama_fullpen": (mk(MALAMA, model=4), TimeSpec(True, -1, 8, 50), [None]),
    "malama_k10": (dict(quiet=0, model=5, dimless=True, timeseries=True, piezometer=True,
                        Q=0.202, l=6.2496, d=2.6001, rw=0.1, rc=0.1, gammaSkin=1.0, timeType=1, timePar=[0.0, 1.0],
                        b=9.0, Kr=6.37e-5, kappa=0.4458, Ss=5.67e-5, Sy=0.301, beta=9.0, MoenchM=0, MoenchAlpha=[],
                        ac=5.68, ak=23.66, psia=0.341, psik=0.3098, usL=2.75, MNtype=2, order=19,
                        M=18, alpha=1.0e-8, tol=1.0e-9, k=10, R=8, j0s=[2, 2], nacc=10, ord=40,
                        tval=5.5, rval=0.25, zTop=4.5, zBot=0.0, zOrd=3, rwobs=0.01, sF=20.0),
                   TimeSpec(True, -1, 8, 30), [None]),
    # model 2: Hantush with wellbore storage and observation-well delay (needs Amos K0,K1)
    "hstorage_fullpen_lay1": (mk(MALAMA, model=2, l=52.669, d=0.0, k=6, R=4, j0s=[1, 1], nacc=10, zTop=0.0, zBot=0.0),
                              TimeSpec(True, -1, 8, 50), [None]),
    "hstorage_fullpen_lay2": (mk(MALAMA, model=2, l=52.669, d=0.0, k=6, R=4, j0s=[1, 1], nacc=10, zTop=30.0, zBot=20.0),
                              TimeSpec(True, -1, 8, 50), [None]),
    "hstorage_partpen_lay1": (mk(MALAMA, model=2, l=40.0, d=10.0, k=6, R=4, j0s=[1, 1], nacc=10, zTop=10.0, zBot=0.0, rw=0.5, rwobs=0.3),
                              TimeSpec(True, -1, 8, 50), [None]),
    "hstorage_partpen_lay2": (mk(MALAMA, model=2, l=40.0, d=10.0, k=6, R=4, j0s=[1, 1], nacc=10, zTop=35.0, zBot=25.0, rw=0.5, rwobs=0.3),
                              TimeSpec(True, -1, 8, 50), [None]),
    # pumping-schedule variants (time.f90:50-80)
    "theis_pulse": (mk(SMALL, model=0, l=7.5, d=2.5, timeType=2, timePar=[0.0, 50.0]), TimeSpec(True, -1, 6, 40), [None]),
    "theis_stairs": (mk(SMALL, model=0, l=7.5, d=2.5, timeType=4, timePar=[100.0, 1000.0]), TimeSpec(True, -1, 6, 40), [None]),
    "theis_square": (mk(SMALL, model=0, l=7.5, d=2.5, timeType=5, timePar=[100.0, 10.0]), TimeSpec(True, -1, 6, 40), [None]),
    "theis_cos": (mk(SMALL, model=0, l=7.5, d=2.5, timeType=6, timePar=[0.01, 10.0]), TimeSpec(True, -1, 6, 40), [None]),
    "theis_sched3": (mk(SMALL, model=0, l=7.5, d=2.5, timeType=-3, timePar=[0.0, 100.0, 1000.0, 5000.0, 1.0, 0.5, 2.0]),
                     TimeSpec(True, -1, 6, 40), [None]),
    "neuman_sched2": (mk(CAPE, timeType=-2, timePar=[0.0, 10.0, 500.0, 1.0, 0.25]), TimeSpec(True, -1, 8, 40), [None]),
    "theis_sqwave": (mk(SMALL, model=0, l=7.5, d=2.5, timeType=8, timePar=[100.0, 10.0]), TimeSpec(True, -1, 6, 40), [None]),
}


def write_decks():
    os.makedirs(DECKS, exist_ok=True)
    for name, (fields, ts, _radii) in DECKSET.items():
        dk = Deck(**fields)
        dk.timeFileName = f"time_{name}.dat"
        dk.spaceFileName = "space_unused.dat"
        dk.outFileName = f"{name}.out"
        dk.write(os.path.join(DECKS, f"{name}.in"))
        ts.write(os.path.join(DECKS, dk.timeFileName))


# ------------------------------------------------------------------ harness driver
def parse_dump(text):
    """returns list of (tag, header-ints, payload-lines)"""
    lines = text.splitlines()
    out = []
    i = 0
    # skip anything read_input printed before the first tag we know
    tags = {"params", "pvalues", "soln", "tanhsinh", "gausslobatto", "wynn", "extrap", "dehoog", "cbesk"}
    while i < len(lines):
        parts = lines[i].split()
        if not parts or parts[0] not in tags:
            i += 1
            continue
        tag = parts[0]
        hdr = [int(x) for x in parts[1:]]
        i += 1
        if tag == "params":
            body = []
            while lines[i].strip() != "endparams":
                body.append(lines[i])
                i += 1
            i += 1
            out.append((tag, hdr, body))
        elif tag == "pvalues":
            out.append((tag, hdr, lines[i:i + hdr[0]])); i += hdr[0]
        elif tag == "soln":
            n = hdr[0] * hdr[1]
            out.append((tag, hdr, lines[i:i + n])); i += n
        elif tag in ("tanhsinh", "gausslobatto"):
            n = 2 * hdr[0]
            out.append((tag, hdr, lines[i:i + n])); i += n
        elif tag == "cbesk":
            out.append((tag, hdr, lines[i:i + 2])); i += 2
        else:
            out.append((tag, hdr, lines[i:i + 1])); i += 1
    return out


def cvec(lines):
    return np.array([[unhx(a), unhx(b)] for a, b in (ln.split() for ln in lines)], dtype=np.float64)


def rvec(lines):
    return np.array([unhx(ln.split()[0]) for ln in lines], dtype=np.float64)


def parse_params(body):
    sc, arrays = {}, {}
    i = 0
    while i < len(body):
        parts = body[i].split()
        name = parts[0]
        if name in ("MoenchGamma", "j0z", "sv", "t", "tD", "rD", "zD", "zLay"):
            n = int(parts[1])
            vals = body[i + 1:i + 1 + n]
            if name in ("sv", "zLay"):
                arrays[name] = np.array([int(v) for v in vals], dtype=np.int32)
            else:
                arrays[name] = rvec(vals)
            i += 1 + n
        else:
            v = parts[1]
            sc[name] = unhx(v) if len(v) == 16 and not v.isdigit() else (unhx(v) if len(v) == 16 else int(v))
            i += 1
    return sc, arrays


INT_KEYS = {"model", "MNtype", "order", "dimless", "timeseries", "piezometer", "nt", "nr", "nz", "zOrd", "M",
            "timeType", "k", "R", "j0s1", "j0s2", "nacc", "ord", "MoenchM"}


def run_harness(name, workdir, flavour="O2"):
    """stage vectors from the reference's public procedures for deck `name`"""
    fields, ts, _ = DECKSET[name]
    dk = Deck(**fields)
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    cmds = ["params"]
    # ---- sample evaluations: 3 times x 9 abscissae (both sides of Re(eta)=MAXEXP, one overflow probe)
    Tc = dk.b ** 2 / (dk.Kr / dk.Ss)
    tlist = [10.0 ** ts.min_log / Tc, 10.0 ** (0.5 * (ts.min_log + ts.max_log)) / Tc, 10.0 ** ts.max_log / Tc]
    rD = dk.rval / dk.b
    alist = [1.0e-3, 0.05, 0.7, 3.1, 7.3, 8.1, 33.0, 250.0, 3000.0]
    soln_in = []
    for tD in tlist:
        cmds.append(f"pvalues {hx(2.0 * tD)}")
        for a in alist:
            aa = a * (1.0 + 0.01 * rng.random())
            cmds.append(f"soln {hx(aa)} {hx(rD)}")
            soln_in.append((tD, aa, rD))
    # ---- quadrature tables for this deck's numerics
    arg = 2.404825557695773 / rD
    for j in range(1, dk.R + 1):
        cmds.append(f"tanhsinh {dk.k - dk.R + j} {hx(arg)}")
    cmds.append(f"gausslobatto {dk.ord}")
    text = subprocess.run([os.path.join(REFBIN[flavour], "ref_harness"), f"{name}.in"], input="\n".join(cmds) + "\n",
                          capture_output=True, text=True, cwd=workdir, check=True).stdout
    recs = parse_dump(text)
    it = iter(recs)
    tag, _, body = next(it)
    assert tag == "params"
    sc, arrays = parse_params(body)
    arrs = {f"par_{k}": v for k, v in arrays.items()}
    meta = {"deck": name, "scalars_int": {k: int(v) for k, v in sc.items() if k in INT_KEYS},
            "scalars_hex": {k: hx(v) for k, v in sc.items() if k not in INT_KEYS}}
    pv, so = [], []
    k = 0
    for tD in tlist:
        tag, hdr, body = next(it); assert tag == "pvalues"
        pv.append(cvec(body))
        for a in alist:
            tag, hdr, body = next(it); assert tag == "soln", tag
            so.append(cvec(body).reshape(hdr[1], hdr[0], 2))   # [nz][np][2]
            k += 1
    arrs["pv_tee"] = np.array([2.0 * t for t in tlist])
    arrs["pv_p"] = np.stack(pv)                                  # [3][np][2]
    arrs["soln_tD"] = np.array([s[0] for s in soln_in])
    arrs["soln_a"] = np.array([s[1] for s in soln_in])
    arrs["soln_rD"] = np.array([s[2] for s in soln_in])
    arrs["soln_fp"] = np.stack(so)                               # [27][nz][np][2]
    arrs["ts_arg"] = np.array([arg])
    for j in range(1, dk.R + 1):
        tag, hdr, body = next(it); assert tag == "tanhsinh"
        n = hdr[0]
        arrs[f"ts_w{j}"] = rvec(body[:n])
        arrs[f"ts_a{j}"] = rvec(body[n:])
    tag, hdr, body = next(it); assert tag == "gausslobatto"
    n = hdr[0]
    arrs["gl_x"] = rvec(body[:n]); arrs["gl_w"] = rvec(body[n:])
    np.savez_compressed(os.path.join(GOLD, f"stages_{name}.npz"), **arrs)
    with open(os.path.join(GOLD, f"stages_{name}.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def run_generic_stages(workdir, flavour="O2"):
    """wynn_epsilon / extraptozero / deHoog_invlap known-answer vectors (deck independent)"""
    name = "neuman74_partpen"
    rng = np.random.default_rng(12345)
    cmds = []
    wy_in, ex_in, dh_in = [], [], []
    nan, inf = float("nan"), float("inf")

    def series(n, kind):
        j = np.arange(n)
        if kind == "alt":
            s = (-1.0) ** j / (j + 1.0) ** 1.5 * (1.0 + 0.3j) * np.exp(0.2j * j)
        elif kind == "geom":
            s = 0.5 * (-0.7 + 0.2j) ** j
        elif kind == "tiny":
            s = 1e-3 * (-0.5) ** j * (1 + 1j); s[5:] = 1e-19 * (-1.0) ** j[5:]
        elif kind == "const0":
            s = np.zeros(n, complex); s[0] = 1.0
        else:
            s = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.5 ** j
        return s
    cases = []
    for n in (10, 12, 7, 4, 5):
        for kind in ("alt", "geom", "rand"):
            cases.append(series(n, kind))
    cases.append(series(10, "tiny"))
    cases.append(series(10, "const0"))
    s = series(10, "alt"); s[2] = complex(nan, 0.0); cases.append(s)       # NaN at term 3 -> sentinel
    s = series(10, "alt"); s[6] = complex(0.0, nan); cases.append(s)       # NaN at term 7 -> truncate to 6
    s = series(10, "alt"); s[5] = complex(inf, 0.0); cases.append(s)       # Inf at term 6 -> truncate to 5 (odd)
    s = series(10, "alt"); s[4] = complex(nan, nan); cases.append(s)       # truncate to 4
    s = series(10, "geom"); s[0] = complex(nan, 0.0); cases.append(s)      # first term bad -> sentinel
    for s in cases:
        cmds.append(f"wynn {len(s)}")
        cmds += [f"{hx(z.real)} {hx(z.imag)}" for z in s]
        wy_in.append(s)
    for R in (2, 3, 4, 5, 7, 8):
        x = 4.0 / 2.0 ** np.arange(3, 3 + R)
        for _ in range(3):
            y = (1.0 + 0.5j) + (rng.standard_normal(R) + 1j * rng.standard_normal(R)) * x ** 2
            cmds.append(f"extrap {R}")
            cmds += [hx(v) for v in x]
            cmds += [f"{hx(z.real)} {hx(z.imag)}" for z in y]
            ex_in.append((x, y))
    # de Hoog: transforms with known inverses, and degenerate vectors
    for (M, alpha, tol) in ((26, 1e-8, 1e-9), (10, 1e-8, 1e-9), (15, 1e-7, 1e-8), (18, 1e-8, 1e-9), (2, 0.0, 1e-6)):
        cmds.append(f"setlap {M} {hx(alpha)} {hx(tol)}")
        for t in (0.013, 1.0, 37.5, 2.0e5):
            tee = 2.0 * t
            sigma = alpha - np.log(tol) / (2.0 * tee)
            p = sigma + 1j * np.pi * np.arange(2 * M + 1) / tee
            for kind in ("exp", "theis", "nan", "zero"):
                if kind == "exp":
                    fp = 1.0 / (p + 1.0 / t)
                elif kind == "theis":
                    fp = 2.0 / (p * (p + 0.3)) * np.exp(-0.1 * np.sqrt(p))
                elif kind == "nan":
                    fp = 1.0 / (p + 1.0 / t); fp[3] = complex(nan, 0.0); fp[2 * M] = complex(0.0, nan)
                else:
                    fp = np.zeros_like(p)
                cmds.append(f"dehoog {hx(t)} {hx(tee)}")
                cmds += [f"{hx(z.real)} {hx(z.imag)}" for z in fp]
                dh_in.append((M, alpha, tol, t, tee, fp))
    # Amos K0/K1 (cbessel.f90:877): both branches (|z| <= 2 series, > 2 Miller), right half plane
    kz = []
    for mag in (1e-12, 1e-6, 1e-3, 0.05, 0.7, 1.9, 2.0, 2.0000001, 2.7, 5.0, 12.0, 28.0, 29.0, 60.0, 300.0):
        for ang in (0.0, 0.3, 0.78, 1.2, 1.5, -0.6, -1.45):
            z = mag * np.exp(1j * ang)
            kz.append(z)
            cmds.append(f"cbesk {hx(z.real)} {hx(z.imag)}")
    text = subprocess.run([os.path.join(REFBIN[flavour], "ref_harness"), f"{name}.in"], input="\n".join(cmds) + "\n",
                          capture_output=True, text=True, cwd=workdir, check=True).stdout
    recs = [r for r in parse_dump(text)]
    it = iter(recs)
    arrs = {}
    for i, s in enumerate(wy_in):
        tag, _, body = next(it); assert tag == "wynn"
        arrs[f"wynn_in_{i}"] = np.stack([s.real, s.imag], -1)
        arrs[f"wynn_out_{i}"] = cvec(body)[0]
    for i, (x, y) in enumerate(ex_in):
        tag, _, body = next(it); assert tag == "extrap"
        arrs[f"extrap_x_{i}"] = x
        arrs[f"extrap_y_{i}"] = np.stack([y.real, y.imag], -1)
        arrs[f"extrap_out_{i}"] = cvec(body)[0]
    for i, (M, alpha, tol, t, tee, fp) in enumerate(dh_in):
        tag, _, body = next(it); assert tag == "dehoog"
        arrs[f"dehoog_par_{i}"] = np.array([M, alpha, tol, t, tee])
        arrs[f"dehoog_fp_{i}"] = np.stack([fp.real, fp.imag], -1)
        arrs[f"dehoog_out_{i}"] = rvec(body)
    kout, kerr = [], []
    for z in kz:
        tag, hdr, body = next(it); assert tag == "cbesk"
        kerr.append(hdr)
        kout.append(cvec(body))
    arrs["cbesk_z"] = np.array([[z.real, z.imag] for z in kz])
    arrs["cbesk_k"] = np.stack(kout)                 # [n][2][2]: K0, K1
    arrs["cbesk_nz_ierr"] = np.array(kerr)
    arrs["counts"] = np.array([len(wy_in), len(ex_in), len(dh_in)])
    np.savez_compressed(os.path.join(GOLD, "stages_generic.npz"), **arrs)


# ------------------------------------------------------------------ end-to-end
def parse_out(path):
    rows = []
    with open(path, errors="replace") as f:     # the reference's header may carry raw bytes
        for ln in f:
            if ln.startswith("#") or not ln.strip():
                continue
            rows.append([float(x) for x in ln.split()[:3]])
    return np.array(rows)


def run_e2e(name, workdir, flavours=("O2", "O3native"), threads=8):
    fields, ts, radii = DECKSET[name]
    arrs = {}
    rlist = []
    for ir, r in enumerate(radii):
        dk = Deck(**fields)
        if r is not None:
            dk.rval = r
        rlist.append(dk.rval)
        dk.timeFileName = f"time_{name}.dat"
        dk.spaceFileName = "space_unused.dat"
        dk.outFileName = f"{name}_{ir}.out"
        dk.write(os.path.join(workdir, f"{name}_{ir}.in"))
        for fl in flavours:
            env = dict(os.environ, OMP_NUM_THREADS=str(threads))
            subprocess.run([os.path.join(REFBIN[fl], "unconfined"), f"{name}_{ir}.in"], cwd=workdir, env=env,
                           check=True, capture_output=True)
            out = parse_out(os.path.join(workdir, dk.outFileName))
            arrs[f"{fl}_r{ir}"] = out           # columns: t (or tD), h, dh
            print(f"  e2e {name} r={dk.rval} {fl}: {out.shape[0]} rows", flush=True)
    arrs["radii"] = np.array(rlist)
    np.savez_compressed(os.path.join(GOLD, f"e2e_{name}.npz"), **arrs)


def main():
    which = sys.argv[1:] or ["decks", "stages", "generic", "e2e"]
    write_decks()
    work = tempfile.mkdtemp(prefix="ucf_gold_")
    try:
        for fn in os.listdir(DECKS):
            shutil.copy(os.path.join(DECKS, fn), work)
        names = [n for n in DECKSET if not any(w.startswith("only=") for w in which) or
                 n in [w[5:] for w in which if w.startswith("only=")]]
        if "stages" in which:
            for name in names:
                print("stages", name, flush=True)
                run_harness(name, work)
        if "generic" in which:
            run_generic_stages(work)
        if "e2e" in which:
            for name in names:
                print("e2e", name, flush=True)
                run_e2e(name, work)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
