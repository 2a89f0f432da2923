"""Input deck of the well-test simulator: the 18-line free-format text file plus
its time / space companion files.

Mirrors what the reference's ``read_input`` accepts (reference
driver_io.f90:88-528, format described in input-explanation.txt:1-288): list-
directed reads, i.e. the first N whitespace separated tokens of each line are
consumed and the rest of the line is a comment.  Only parsing lives here; all
arithmetic on the values (non-dimensionalisation, J0 zeros, split vector) is
done by the native library (``ucf_plan_create`` and friends) so that it is
bit-identical with the reference's.
"""
from __future__ import annotations

import dataclasses
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence


class DeckError(ValueError):
    """Raised where the reference prints ``ERROR ...`` and stops."""


def _fnum(tok: str) -> float:
    t = tok.strip().rstrip(",")
    t = t.replace("D", "E").replace("d", "e")
    return float(t)


def _inum(tok: str) -> int:
    t = tok.strip().rstrip(",")
    try:
        return int(t)
    except ValueError as exc:  # the reference aborts: "Bad character in INTEGER input field"
        raise DeckError(f"bad integer field {tok!r}") from exc


def _lnum(tok: str) -> bool:
    t = tok.strip().lstrip(".").upper()
    if t.startswith("T"):
        return True
    if t.startswith("F"):
        return False
    raise DeckError(f"bad logical field {tok!r}")


def _toks(line: str, n: int, what: str) -> List[str]:
    parts = line.split()
    if len(parts) < n:
        raise DeckError(f"deck line for {what}: expected {n} values, found {len(parts)}")
    return parts[:n]


def _ffmt(x: float) -> str:
    """shortest text that reads back to the same binary64 (Fortran list-directed safe)"""
    s = repr(float(x))
    if "e" in s or "E" in s:
        return s.replace("e", "D")
    return s + "D0"


@dataclass
class TimeSpec:
    """time file (driver_io.f90:414-452): computed log-spaced vector or explicit list"""
    compute: bool = True
    min_log: int = -1
    max_log: int = 8
    n: int = 100
    times: Optional[List[float]] = None

    def write(self, path: str) -> None:
        with open(path, "w") as f:
            nfile = len(self.times) if self.times else 0
            f.write(f"{'T' if self.compute else 'F'}  {nfile}   :: compute times?, # times listed below\n")
            f.write(f"{self.min_log}  {self.max_log}  {self.n}   :: log10(tmin), log10(tmax), # times\n")
            for t in self.times or []:
                f.write(_ffmt(t) + "\n")

    @staticmethod
    def read(path: str) -> "TimeSpec":
        with open(path) as f:
            lines = f.read().splitlines()
        a = _toks(lines[0], 2, "time file line 1")
        b = _toks(lines[1], 3, "time file line 2")
        ts = TimeSpec(_lnum(a[0]), _inum(b[0]), _inum(b[1]), _inum(b[2]))
        nfile = _inum(a[1])
        if not ts.compute:
            ts.times = [_fnum(_toks(lines[2 + i], 1, "time value")[0]) for i in range(nfile)]
        return ts


@dataclass
class SpaceSpec:
    """space file (driver_io.f90:470-523): computed linear grids or explicit lists"""
    compute: bool = True
    min_r: float = 1.0
    max_r: float = 10.0
    n_r: int = 2
    min_z: float = 0.0
    max_z: float = 1.0
    n_z: int = 1
    r: Optional[List[float]] = None
    z: Optional[List[float]] = None

    def write(self, path: str) -> None:
        with open(path, "w") as f:
            nr = len(self.r) if self.r else 0
            nz = len(self.z) if self.z else 0
            f.write(f"{'T' if self.compute else 'F'}  {nr}  {nz}   :: compute locations?, # r, # z listed below\n")
            f.write(f"{_ffmt(self.min_r)}  {_ffmt(self.max_r)}  {self.n_r}   :: rmin, rmax, # r\n")
            f.write(f"{_ffmt(self.min_z)}  {_ffmt(self.max_z)}  {self.n_z}   :: zmin, zmax, # z\n")
            f.write(" ".join(_ffmt(x) for x in (self.r or [])) + "\n")
            f.write(" ".join(_ffmt(x) for x in (self.z or [])) + "\n")

    @staticmethod
    def read(path: str) -> "SpaceSpec":
        with open(path) as f:
            lines = f.read().splitlines()
        a = _toks(lines[0], 3, "space file line 1")
        b = _toks(lines[1], 3, "space file line 2")
        c = _toks(lines[2], 3, "space file line 3")
        sp = SpaceSpec(_lnum(a[0]), _fnum(b[0]), _fnum(b[1]), _inum(b[2]), _fnum(c[0]), _fnum(c[1]), _inum(c[2]))
        nr, nz = _inum(a[1]), _inum(a[2])
        if not sp.compute:
            sp.r = [_fnum(t) for t in _toks(lines[3], nr, "r values")]
            sp.z = [_fnum(t) for t in _toks(lines[4], nz, "z values")]
        return sp


@dataclass
class Deck:
    # line 1 (driver_io.f90:88)
    quiet: int = 0
    model: int = 5
    dimless: bool = False
    timeseries: bool = True
    piezometer: bool = True
    # lines 2-6 (driver_io.f90:102-127)
    Q: float = 1.0
    l: float = 1.0
    d: float = 0.0
    rw: float = 0.1
    rc: float = 0.1
    gammaSkin: float = 1.0
    timeType: int = 1
    timePar: List[float] = field(default_factory=lambda: [0.0, 1.0])
    # lines 7-11 (driver_io.f90:130-157)
    b: float = 1.0
    Kr: float = 1.0
    kappa: float = 1.0
    Ss: float = 1.0e-4
    Sy: float = 0.2
    beta: float = 0.0
    MoenchM: int = 0
    MoenchAlpha: List[float] = field(default_factory=list)
    ac: float = 1.0
    ak: float = 1.0
    psia: float = 0.0
    psik: float = 0.0
    usL: float = 1.0
    MNtype: int = 2
    order: int = 5
    # lines 12-14 (driver_io.f90:296-304)
    M: int = 26
    alpha: float = 1.0e-8
    tol: float = 1.0e-9
    k: int = 6
    R: int = 4
    j0s: List[int] = field(default_factory=lambda: [1, 1])
    nacc: int = 10
    ord: int = 50
    # lines 15-18 (driver_io.f90:341-351,527)
    timeFileName: str = "timedata.dat"
    tval: float = 1.0
    spaceFileName: str = "spacedata.dat"
    rval: float = 1.0
    zTop: float = 1.0
    zBot: float = 0.0
    zOrd: int = 1
    rwobs: float = 0.1
    sF: float = 1.0
    outFileName: str = "ucf.out"

    # ------------------------------------------------------------------ text
    @staticmethod
    def parse(text: str) -> "Deck":
        lines = [ln for ln in text.splitlines()]
        if len(lines) < 18:
            raise DeckError(f"deck has {len(lines)} lines, 18 required")
        dk = Deck()
        t = _toks(lines[0], 5, "switches")
        dk.quiet, dk.model = _inum(t[0]), _inum(t[1])
        dk.dimless, dk.timeseries, dk.piezometer = _lnum(t[2]), _lnum(t[3]), _lnum(t[4])
        dk.Q = _fnum(_toks(lines[1], 1, "Q")[0])
        t = _toks(lines[2], 2, "l,d"); dk.l, dk.d = _fnum(t[0]), _fnum(t[1])
        t = _toks(lines[3], 2, "rw,rc"); dk.rw, dk.rc = _fnum(t[0]), _fnum(t[1])
        dk.gammaSkin = _fnum(_toks(lines[4], 1, "gamma")[0])
        dk.timeType = _inum(_toks(lines[5], 1, "time behaviour")[0])
        # -n: n-step piecewise constant (n = 1..100); -(100+n): n-segment piecewise linear; 2n+1 parameters either way
        # (the reference sizes the record with mod(type,100), driver_io.f90:124, which breaks for -100 and -200)
        if dk.timeType > -1:
            npar = 2
        else:
            nseg = -dk.timeType if dk.timeType >= -100 else -dk.timeType - 100
            npar = 2 * nseg + 1
        t = _toks(lines[5], 1 + npar, "time behaviour parameters")
        dk.timePar = [_fnum(x) for x in t[1:]]
        dk.b = _fnum(_toks(lines[6], 1, "b")[0])
        t = _toks(lines[7], 2, "Kr,kappa"); dk.Kr, dk.kappa = _fnum(t[0]), _fnum(t[1])
        t = _toks(lines[8], 2, "Ss,Sy"); dk.Ss, dk.Sy = _fnum(t[0]), _fnum(t[1])
        t = _toks(lines[9], 2, "beta, MoenchM"); dk.beta, dk.MoenchM = _fnum(t[0]), _inum(t[1])
        if dk.MoenchM > 0:
            t = _toks(lines[9], 2 + dk.MoenchM, "Moench alphas")
            dk.MoenchAlpha = [_fnum(x) for x in t[2:]]
        else:
            dk.MoenchAlpha = []
        t = _toks(lines[10], 7, "Mishra/Neuman parameters")
        dk.ac, dk.ak, dk.psia, dk.psik, dk.usL = (_fnum(x) for x in t[:5])
        dk.MNtype, dk.order = _inum(t[5]), _inum(t[6])
        t = _toks(lines[11], 3, "deHoog"); dk.M, dk.alpha, dk.tol = _inum(t[0]), _fnum(t[1]), _fnum(t[2])
        t = _toks(lines[12], 2, "tanh-sinh"); dk.k, dk.R = _inum(t[0]), _inum(t[1])
        t = _toks(lines[13], 4, "Gauss-Lobatto")
        dk.j0s = [_inum(t[0]), _inum(t[1])]; dk.nacc, dk.ord = _inum(t[2]), _inum(t[3])
        t = _toks(lines[14], 2, "time file"); dk.timeFileName, dk.tval = t[0], _fnum(t[1])
        t = _toks(lines[15], 2, "space file"); dk.spaceFileName, dk.rval = t[0], _fnum(t[1])
        t = _toks(lines[16], 5, "observation well")
        dk.zTop, dk.zBot, dk.zOrd, dk.rwobs, dk.sF = _fnum(t[0]), _fnum(t[1]), _inum(t[2]), _fnum(t[3]), _fnum(t[4])
        dk.outFileName = _toks(lines[17], 1, "output file")[0]
        return dk

    @staticmethod
    def read(path: str) -> "Deck":
        with open(path) as f:
            return Deck.parse(f.read())

    def text(self) -> str:
        L = lambda b: "T" if b else "F"
        alphas = " ".join(_ffmt(a) for a in self.MoenchAlpha) if self.MoenchM > 0 else "-999."
        rows = [
            (f"{self.quiet}  {self.model}  {L(self.dimless)}  {L(self.timeseries)}  {L(self.piezometer)}",
             "verbosity, model 0-6, dimensionless out?, time series?, piezometer?"),
            (_ffmt(self.Q), "Q pumping rate"),
            (f"{_ffmt(self.l)}  {_ffmt(self.d)}", "l, d: depth below aquifer top of screen bottom, top"),
            (f"{_ffmt(self.rw)}  {_ffmt(self.rc)}", "rw, rc"),
            (_ffmt(self.gammaSkin), "skin"),
            (f"{self.timeType}  " + "  ".join(_ffmt(x) for x in self.timePar), "pumping time behaviour, parameters"),
            (_ffmt(self.b), "b saturated thickness"),
            (f"{_ffmt(self.Kr)}  {_ffmt(self.kappa)}", "Kr, kappa=Kz/Kr"),
            (f"{_ffmt(self.Ss)}  {_ffmt(self.Sy)}", "Ss, Sy"),
            (f"{_ffmt(self.beta)}  {self.MoenchM}  {alphas}", "Malama beta, # Moench alphas, alphas"),
            (f"{_ffmt(self.ac)}  {_ffmt(self.ak)}  {_ffmt(self.psia)}  {_ffmt(self.psik)}  {_ffmt(self.usL)}  {self.MNtype}  {self.order}",
             "Mishra/Neuman a_c, a_k, psi_a, psi_k, L, type, FD order"),
            (f"{self.M}  {_ffmt(self.alpha)}  {_ffmt(self.tol)}", "de Hoog M, alpha, tol"),
            (f"{self.k}  {self.R}", "tanh-sinh k, Richardson levels"),
            (f"{self.j0s[0]}  {self.j0s[1]}  {self.nacc}  {self.ord}", "J0 split min/max, # zeros accelerated, GL order"),
            (f"{self.timeFileName}  {_ffmt(self.tval)}", "time file, t (contour mode)"),
            (f"{self.spaceFileName}  {_ffmt(self.rval)}", "space file, r (time-series mode)"),
            (f"{_ffmt(self.zTop)}  {_ffmt(self.zBot)}  {self.zOrd}  {_ffmt(self.rwobs)}  {_ffmt(self.sF)}",
             "obs screen top, bottom (z up from aquifer base), # z points, obs radius, shape factor"),
            (self.outFileName, ""),
        ]
        return "\n".join(f"{a:<64s}" + (f" :: {c}" if c else "") for a, c in rows) + "\n"

    def write(self, path: str) -> None:
        with open(path, "w") as f:
            f.write(self.text())

    def replace(self, **kw) -> "Deck":
        return dataclasses.replace(self, **kw)

    def check_observation(self, t, r, z, space_computed: bool = True) -> None:
        """the checks on where and when the solution is observed, on which the reference prints ERROR and stops
        (driver_io.f90:355-394 time series, :443-449 listed times, :494-519 contour grids)"""
        if self.timeseries:
            if self.zTop < self.zBot:
                raise DeckError(f"for screened observation wells top of monitoring well screen must be at or above bottom: "
                                f"top={self.zTop} bot={self.zBot}")
            if self.zTop > self.b or self.zBot < 0.0:
                raise DeckError(f"top of monitoring well screen must be above bottom and both between 0 and b: "
                                f"top={self.zTop} bot={self.zBot} b={self.b}")
            if not self.piezometer and self.zOrd < 1:
                raise DeckError(f"# of quadrature points at monitoring location must be > 0: {self.zOrd}")
            if self.rwobs <= 0.0:
                raise DeckError(f"monitoring well radius must be >0: {self.rwobs}")
            if self.sF <= 0.0:
                raise DeckError(f"monitoring well shape factor must be >0: {self.sF}")
            if not self.rval > self.rw:
                raise DeckError(f"r must be > rw: r={self.rval} rw={self.rw}")
            if any(x < 0.0 for x in t):
                raise DeckError("all times must be > 0")
        else:
            if any(x < 0.0 or x > self.b for x in z):
                raise DeckError(f"z must be in range 0<=>b: zmin={min(z)} zmax={max(z)} b={self.b}")
            if not space_computed and any(x < self.rw for x in r):        # (the reference checks listed radii only)
                raise DeckError(f"r must be >= rw: rmin={min(r)} rw={self.rw}")


def resolve(deck_path: str, name: str) -> str:
    """companion files are opened relative to the working directory in the reference;
    here: relative to the deck's directory if not found in the cwd"""
    if os.path.exists(name):
        return name
    cand = os.path.join(os.path.dirname(os.path.abspath(deck_path)), name)
    return cand
