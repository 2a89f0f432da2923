"""Output files in the reference's format, so that results can be diffed line by line with
`./unconfined deck`'s: the '#' header echo (reference driver_io.f90:668-845) and the result rows
(driver.f90:245-273) with the edit descriptors of constants.f90:72-74 (ES14.07E2, ES24.15E4).

Pure formatting -- no numerics.  tests/test_output_format.py checks byte equality against files
written by the reference binary itself.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Sequence

MODEL_DESCRIP = ["Theis", "Hantush", "Hantush w/ stor", "Moench", "Malama full pen", "Malama part pen", "Mishra/Neuman"]
# types.f90:66-75 (text is part of the header line of every output file)
TIME_DESCRIP = [
    "step on; tpar(1) = on time; tpar(2) not used",
    "finite pulse; tpar(1:2) = on/off time",
    "infinitessimal pulse; tpar(1) = pulse location; tpar(2) not used",
    "stairs; tpar(1) = time step (Q increase by integer multiples); tpar(2) = off time",
    "rectified square wave; tpar(1) = 1/2 period of wave; tpar(2) = start time",
    "cos(omega*t); tpar(1) = omega; tpar(2) = start time",
    "rectified triangular wave; tpar(1) = 1/4 period of wave; tpar(2) = start time",
    "rectified square wave; tpar(1) = 1/2 period of wave; tpar(2) = start time",
    "piecewise constant rate (n steps); tpar(1:n)=ti; tpar(n+1)=tfinal; tpar(n+2:)=Q",
]


def es(x: float, width: int, digits: int, expw: int) -> str:
    """Fortran ESw.dEe"""
    if math.isnan(x):
        return "NaN".rjust(width)
    if math.isinf(x):
        return ("Inf" if x > 0 else "-Inf").rjust(width)
    s = f"{x:.{digits}E}"
    mant, exp = s.split("E")
    sign, ev = exp[0], exp[1:].lstrip("0") or "0"
    if len(ev) > expw:
        return "*" * width
    out = f"{mant}E{sign}{ev.rjust(expw, '0')}"
    return out.rjust(width) if len(out) <= width else "*" * width


def rfmt(x: float) -> str:      # RFMT = ES14.07E2
    return es(x, 14, 7, 2)


def hfmt(x: float) -> str:      # HFMT = ES24.15E4
    return es(x, 24, 15, 4)


def _l(b: bool) -> str:
    return "T" if b else "F"


def _common_head(dk, tag: str) -> List[str]:
    tp = list(dk.timePar)
    tdesc = TIME_DESCRIP[dk.timeType - 1] if dk.timeType > 0 else TIME_DESCRIP[8]
    return [
        f"# Q (volumetric pumping rate) :: {rfmt(dk.Q)}",
        f"# b (initial sat thickness) :: {rfmt(dk.b)}",
        "# l,d (screen bot & top) :: " + " ".join(rfmt(v) for v in (dk.l, dk.d)),
        "# rw,rc (well/casing radii) :: " + " ".join(rfmt(v) for v in (dk.rw, dk.rc)),
        ("# Kr,kappa (kappa=Kz/Kr) :: " if tag == "ts" else "# Kr,kappa (Kz/Kr) :: ") + " ".join(rfmt(v) for v in (dk.Kr, dk.kappa)),
        "# Ss,Sy :: " + " ".join(rfmt(v) for v in (dk.Ss, dk.Sy)),
        ("# gamma (dimensionless skin) :: " if tag == "ts" else "# gamma (dimless skin) :: ") + rfmt(dk.gammaSkin),
        f"# pumping well time behavior :: {dk.timeType}{tdesc}" + " ".join(rfmt(v) for v in tp),
        f"# deHoog M, alpha, tol :: {dk.M}" + " ".join(rfmt(v) for v in (dk.alpha, dk.tol)),
        f"# tanh-sinh: k, n extrapolation steps :: {dk.k} {dk.R}",
        f"# GLquad: J0 split, n 0-accel, GL-order :: {dk.j0s[0]} {dk.j0s[1]} {dk.nacc} {dk.ord}",
    ]


def _model_lines(dk, D, tag: str) -> List[str]:
    out = []
    if dk.model in (4, 5):
        out.append(f"# Malama beta linearization parameter :: {rfmt(dk.beta)}")
    elif dk.model == 6:
        out.append("# Mishra/Neuman ac,ak,psia,psik,b1 ::" + " ".join(rfmt(v) for v in (D.ac_eff, dk.ak, dk.psia, dk.psik, D.b1)))
        if dk.MNtype == 2:
            lab = ("# Mishra/Neuman vadose zone finite-difference order, finite-difference spacing ::" if tag == "ts"
                   else "# Mishra/Neuman finite-difference order, finite-difference mesh spacing ::")
            out.append(f"{lab}{dk.order} {rfmt(dk.usL / (dk.order - 1))}")
        elif dk.MNtype == 1 and tag == "ts":
            out.append("# NB: Malama's Mishra/Neuman implementation (1) assumes ac=ak and fully penetrating pumping "
                       "well without wellbore storage")
    # model 3: the reference sends its Moench line to stdout, not to the file (quirk Q7)
    return out


def timeseries_header(dk, D, r: float, rD: float, z0: float, zD0: float, nt: int, ep_kind: int = 8) -> List[str]:
    """driver_io.f90:668-757"""
    L = ["# -*-auto-revert-*-",
         f"# model, EP precision :: {dk.model} {MODEL_DESCRIP[dk.model]}, {ep_kind}",
         f"# dimensionless?, timeseries?, piezometer? :: {_l(dk.dimless)} {_l(dk.timeseries)} {_l(dk.piezometer)}"]
    L += _common_head(dk, "ts")
    if dk.piezometer:
        L.append("# point obs piezometer r,rD,z,zD :: " + " ".join(rfmt(v) for v in (r, rD, z0, zD0)))
    else:
        L.append("# screened obs well r,zTop,zBot,zOrd :: " + " ".join(rfmt(v) for v in (r, dk.zTop, dk.zBot)) + f" {dk.zOrd}")
        L.append("# screened obs well rW,shape factor :: " + " ".join(rfmt(v) for v in (dk.rwobs, dk.sF)))
    L += _model_lines(dk, D, "ts")
    L.append(f"# times :: {nt}")
    L.append("# characteristic length, time :: " + " ".join(rfmt(v) for v in (D.Lc, D.Tc)))
    name = MODEL_DESCRIP[dk.model]
    if dk.dimless:
        L += ["#", "#     t_D              " + name + "             t*dh/d(log(t))"]
    else:
        L.append(f"# characteristic head ::{rfmt(D.Hc)}")
        L += ["#", "#     t                " + name + "             t*dh/d(log(t))"]
    L.append("#---------------------------------------------------------------")
    return L


def contour_header(dk, D, r: Sequence[float], z: Sequence[float], t: float, tD: float, ep_kind: int = 8) -> List[str]:
    """driver_io.f90:760-845"""
    L = ["# -*-auto-revert-*-",
         f"# model, EP :: {dk.model} {MODEL_DESCRIP[dk.model]}, {ep_kind}",
         f"# dimensionless?, timeseries? :: {_l(dk.dimless)} {_l(dk.timeseries)}"]
    L += _common_head(dk, "ct")
    L.append(f"# num r locations, rlocs :: {len(r)} " + " ".join(rfmt(v) for v in r))
    L.append(f"# num z locations, zlocs :: {len(z)} " + " ".join(rfmt(v) for v in z))
    L.append("# time, tD :: " + " ".join(rfmt(v) for v in (t, tD)))
    L += _model_lines(dk, D, "ct")
    name = MODEL_DESCRIP[dk.model]
    L.append("#")
    if dk.dimless:
        L.append("#     z_D           r_D      " + "     " + name + "          t*dh/d(log(t))")
    else:
        L.append("#      z            r        " + "     " + name + "          t*dh/d(log(t))")
    L.append("#----------------------------------------------------------------------------")
    return L


def timeseries_rows(t: Iterable[float], h: Iterable[float], dh: Iterable[float]) -> List[str]:
    """driver.f90:245-252"""
    return [f"{rfmt(a)} {hfmt(b)} {hfmt(c)}" for a, b, c in zip(t, h, dh)]


def contour_rows(z: Sequence[float], r: Sequence[float], h, dh) -> List[str]:
    """driver.f90:259-271; h, dh indexed [ir][iz]"""
    rows = []
    for k, rv in enumerate(r):
        for m, zv in enumerate(z):
            rows.append(f"{rfmt(zv)} {rfmt(rv)} {hfmt(h[k][m])} {hfmt(dh[k][m])}")
    return rows


def write_lines(path: str, lines: Iterable[str]) -> None:
    with open(path, "w") as f:
        for ln in lines:
            f.write(ln + "\n")
