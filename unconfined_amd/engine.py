"""Host-side mirror of the reference's operator interface for the drawdown path.

The reference has no library API; its de-facto interface is the set of module
procedures ``program Driver`` imports (reference driver.f90:28-40).  The same
names are offered here on top of the C ABI:

    Plan(params)                       read_input's numerical half + first-time setup
    Plan.drawdown(tD, rD, sv, zD, zLay)  the (i,k) loop body, driver.f90:100-232
    Plan.lap_hank_soln / pvalues / ...  the imported procedures, for stage parity
    run_deck(path)                     program Driver for one deck (time-series or contour)

Everything numerical happens in libucf.so on the GPU (or, for the once-per-run
table setup, in its host code); this module only marshals arrays.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

from . import lib as _libmod
from .abi import UcfDerived, UcfParams, UcfStats, params_from_deck
from .deck import Deck, SpaceSpec, TimeSpec, resolve
from .host import screen_average_np


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Plan:
    """Immutable per-parameter-set state: dimensionless parameters, J0 zeros,
    tanh-sinh / Gauss-Lobatto tables (resident on the GPU)."""

    def __init__(self, params: UcfParams, mode: str = "faithful", layout: str = "auto"):
        self._lib = _libmod.load()
        self._h = C.c_void_p()
        self.params = params
        _libmod.check(self._lib.ucf_plan_create(C.byref(params), C.byref(self._h)))
        self.derived = UcfDerived()
        _libmod.check(self._lib.ucf_plan_derived(self._h, C.byref(self.derived)))
        self.set_mode(mode, layout)

    @classmethod
    def from_deck(cls, dk: Deck, mode: str = "faithful") -> "Plan":
        return cls(params_from_deck(dk), mode)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ucf_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mode(self, mode: str, layout: str = "auto"):
        """mode: 'faithful' | 'fast'; layout (grids only): 'auto' picks lane = time when that fills the
        wave better, 'sample' forces lane = Laplace sample (diagnostic / A-B timing)"""
        m = {"faithful": 0, "fast": 1}[mode] | ({"auto": 0, "sample": 2}[layout])
        _libmod.check(self._lib.ucf_plan_set_mode(self._h, m))
        self.mode = mode
        self.layout = layout

    def set_timing(self, enable: bool = True):
        _libmod.check(self._lib.ucf_plan_set_timing(self._h, 1 if enable else 0))

    def kernel_ms(self):
        """(duration in ms, kernel name) of the dominant kernel of the last timed grid call"""
        ms = C.c_double(0.0)
        name = C.c_char_p()
        _libmod.check(self._lib.ucf_plan_kernel_ms(self._h, C.byref(ms), C.byref(name)))
        return ms.value, (name.value or b"").decode()

    def kernel_times(self):
        """[(kernel name, total ms, launches), ...] of the kernels of the last timed lane = time grid call, in order of
        first launch (a call that walks the radii in chunks launches every kernel once per chunk)"""
        cap = 16
        ms = (C.c_double * cap)()
        cnt = (C.c_int * cap)()
        names = (C.c_char_p * cap)()
        n = C.c_int(0)
        _libmod.check(self._lib.ucf_plan_kernel_times(self._h, cap, ms, cnt, names, C.byref(n)))
        return [((names[i] or b"").decode(), ms[i], cnt[i]) for i in range(n.value)]

    def reserve(self, nt: int = 0, nr: int = 0, npts: int = 0, nz: int = 1, stream: int = 0):
        """size the workspaces of `stream` so that later *_device calls of these sizes allocate nothing"""
        _libmod.check(self._lib.ucf_plan_reserve(self._h, int(nt), int(nr), int(npts), int(nz), stream or None))

    def alloc_count(self) -> int:
        return int(self._lib.ucf_plan_alloc_count(self._h))

    # ---- tables (read back for parity tests / headers)
    def j0z(self) -> np.ndarray:
        out = np.zeros(self.derived.nj0z)
        _libmod.check(self._lib.ucf_plan_j0z(self._h, len(out), out))
        return out

    def tanh_sinh(self, level: int) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        P = self.params
        n = 2 ** (P.k - P.R + level) - 1
        w = np.zeros(n)
        x = np.zeros(n) if level == P.R else None
        _libmod.check(self._lib.ucf_plan_tanh_sinh(self._h, level, n, w, x.ctypes.data if x is not None else None))
        return w, x

    def gauss_lobatto(self):
        n = self.params.ord - 2
        x, w = np.zeros(n), np.zeros(n)
        _libmod.check(self._lib.ucf_plan_gauss_lobatto(self._h, n, x, w))
        return x, w

    # ---- host helpers
    def zlay(self, zD) -> np.ndarray:
        zD = _f64(zD)
        out = np.zeros(len(zD), np.int32)
        _libmod.check(self._lib.ucf_zlay(self._h, len(zD), zD, out))
        return out

    def split_vector(self, tD) -> np.ndarray:
        tD = _f64(tD)
        out = np.zeros(len(tD), np.int32)
        _libmod.check(self._lib.ucf_split_vector(self._h, len(tD), tD, out))
        return out

    def update(self, params) -> "Plan":
        """new hydraulic / geometric / schedule parameters, same model and numerical settings (ucf_plan_update)"""
        _libmod.check(self._lib.ucf_plan_update(self._h, C.byref(params)))
        self.params = params
        d = UcfDerived()
        _libmod.check(self._lib.ucf_plan_derived(self._h, C.byref(d)))
        self.derived = d
        return self

    def pvalues(self, tee: float) -> np.ndarray:
        out = np.zeros((self.derived.np, 2))
        _libmod.check(self._lib.ucf_pvalues(self._h, float(tee), out))
        return out

    # ---- the hot path
    def drawdown(self, tD, rD, sv, zD, zLay, with_stats: bool = False):
        """h, dh of shape [npts, nz] (dimensionless, before screen averaging)"""
        tD, rD, sv, zD, zLay = _f64(tD), _f64(rD), _i32(sv), _f64(zD), _i32(zLay)
        n, nz = len(tD), len(zD)
        if len(rD) != n or len(sv) != n or len(zLay) != nz:
            raise ValueError("tD, rD, sv must have equal length; zD, zLay too")
        h = np.zeros((n, nz))
        dh = np.zeros((n, nz))
        st = UcfStats()
        _libmod.check(self._lib.ucf_drawdown_batch(self._h, n, tD, rD, sv, nz, zD, zLay, h, dh,
                                                   C.byref(st) if with_stats else None))
        if with_stats:
            return h, dh, {k: getattr(st, k) for k, _ in UcfStats._fields_}
        return h, dh

    def drawdown_grid(self, tD, sv, rD, zD, zLay, with_stats: bool = False):
        """product grid nt x nr (the reference's i/k loop nest): h, dh of shape [nt, nr, nz]"""
        tD, sv, rD, zD, zLay = _f64(tD), _i32(sv), _f64(rD), _f64(zD), _i32(zLay)
        nt, nr, nz = len(tD), len(rD), len(zD)
        if len(sv) != nt or len(zLay) != nz:
            raise ValueError("tD and sv must have equal length; zD and zLay too")
        h = np.zeros((nt, nr, nz))
        dh = np.zeros((nt, nr, nz))
        st = UcfStats()
        _libmod.check(self._lib.ucf_drawdown_grid(self._h, nt, tD, sv, nr, rD, nz, zD, zLay, h, dh,
                                                  C.byref(st) if with_stats else None))
        if with_stats:
            return h, dh, {k: getattr(st, k) for k, _ in UcfStats._fields_}
        return h, dh

    def drawdown_grid_device(self, nt: int, d_tD: int, d_sv: int, nr: int, d_rD: int, zD, zLay, d_h: int, d_dh: int,
                             stream: int = 0, d_stats: int = 0):
        """asynchronous grid launch on device pointers; outputs [nt][nr][nz]"""
        zD, zLay = _f64(zD), _i32(zLay)
        _libmod.check(self._lib.ucf_drawdown_grid_device(self._h, int(nt), d_tD, d_sv, int(nr), d_rD, len(zD), zD, zLay,
                                                         d_h, d_dh, d_stats or None, stream or None))

    def drawdown_grid_shard_device(self, rank: int, world: int, nt: int, d_tD: int, d_sv: int, nr: int, d_rD: int, zD, zLay,
                                   d_h: int, d_dh: int, stream: int = 0, d_stats: int = 0):
        """this rank's rows (shard_rows) of the nt x nr sweep, written at their place in the full-size device arrays"""
        zD, zLay = _f64(zD), _i32(zLay)
        _libmod.check(self._lib.ucf_drawdown_grid_shard_device(self._h, int(rank), int(world), int(nt), d_tD, d_sv, int(nr), d_rD,
                                                               len(zD), zD, zLay, d_h, d_dh, d_stats or None, stream or None))

    def drawdown_grid_allgather(self, comm: int, rank: int, world: int, nt: int, d_tD: int, d_sv: int, nr: int, d_rD: int, zD, zLay,
                                d_h: int, d_dh: int, stream: int = 0, d_stats: int = 0):
        """this rank's rows, then the in-place RCCL all-gather of h and dh over `comm` (an ncclComm_t as an int:
        comm_create() below or the host's own), all on `stream` (ucf_drawdown_grid_allgather)"""
        zD, zLay = _f64(zD), _i32(zLay)
        _libmod.check(self._lib.ucf_drawdown_grid_allgather(self._h, int(rank), int(world), int(nt), d_tD, d_sv, int(nr), d_rD,
                                                            len(zD), zD, zLay, d_h, d_dh, d_stats or None, comm, stream or None))

    def drawdown_device(self, n: int, d_tD: int, d_rD: int, d_sv: int, zD, zLay, d_h: int, d_dh: int,
                        stream: int = 0, d_stats: int = 0):
        """asynchronous launch on device pointers (ints), e.g. torch tensors' data_ptr()"""
        zD, zLay = _f64(zD), _i32(zLay)
        _libmod.check(self._lib.ucf_drawdown_batch_device(self._h, int(n), d_tD, d_rD, d_sv, len(zD), zD, zLay,
                                                          d_h, d_dh, d_stats or None, stream or None))

    # ---- stage hooks
    def debug_stages(self, tD, sv, rD, zD, zLay, grid: bool = True):
        """the intermediate stages of the PRODUCTION launch sequence (ucf_debug_stages): dict with
        state [npts, np, R+1+nacc, nz] complex (level sums without arg/2 | running area | J0-interval areas), ndone [npts, np],
        totlap [npts, nz, np] complex, h, dh [npts, nz], layout.  grid: nt times x nr radii, points in order it*nr + ir;
        else a point list taken as ordered by radius"""
        tD, sv, rD, zD, zLay = _f64(tD), _i32(sv), _f64(rD), _f64(zD), _i32(zLay)
        nt, nr, nz = len(tD), len(rD), len(zD)
        npts = nt * nr if grid else nt
        D = self.derived
        slots = (self.params.R + 1 + self.params.nacc) * nz
        state = np.zeros((npts, D.np, slots, 2))
        ndone = np.zeros((npts, D.np), np.int32)
        totlap = np.zeros((npts, nz, D.np, 2))
        h = np.zeros((npts, nz)); dh = np.zeros((npts, nz))
        info = np.zeros(4, np.int32)
        _libmod.check(self._lib.ucf_debug_stages(self._h, 1 if grid else 0, nt, tD, sv, nr, rD, nz, zD, zLay, state, ndone, totlap, h, dh, info))
        st = (state[..., 0] + 1j * state[..., 1]).reshape(npts, D.np, self.params.R + 1 + self.params.nacc, nz)
        return {"state": st, "has_state": bool(info[1]), "ndone": ndone, "totlap": totlap[..., 0] + 1j * totlap[..., 1], "h": h, "dh": dh,
                "layout": int(info[0]), "R": self.params.R, "nacc": self.params.nacc}

    def lap_hank_soln(self, a, rD: float, p, zD, zLay) -> np.ndarray:
        """fp[n_a, nz, np, 2]"""
        a, p, zD, zLay = _f64(np.atleast_1d(a)), _f64(p), _f64(zD), _i32(zLay)
        out = np.zeros((len(a), len(zD), p.shape[0], 2))
        _libmod.check(self._lib.ucf_eval_samples(self._h, len(a), a, float(rD), p.shape[0], p, len(zD), zD, zLay, out))
        return out


def shard_rows(nt: int, world: int, rank: int) -> Tuple[int, int]:
    """rows [lo, hi) of the nt-row sweep that shard `rank` of `world` owns (ucf_shard_rows; needs no GPU)"""
    lo, hi = C.c_int(0), C.c_int(0)
    _libmod.check(_libmod.load().ucf_shard_rows(int(nt), int(world), int(rank), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def comm_unique_id() -> bytes:
    """128-byte id of a new RCCL communicator (rank 0 draws it and hands it to the other ranks: ucf_comm_unique_id)"""
    buf = C.create_string_buffer(128)
    _libmod.check(_libmod.load().ucf_comm_unique_id(buf))
    return buf.raw


def comm_create(unique_id: bytes, world: int, rank: int) -> int:
    """RCCL communicator of the library's own on the current HIP device (ucf_comm_create); returns the handle"""
    if len(unique_id) != 128:
        raise ValueError("the unique id has 128 bytes")
    h = C.c_void_p()
    _libmod.check(_libmod.load().ucf_comm_create(unique_id, int(world), int(rank), C.byref(h)))
    return h.value


def comm_destroy(comm: int) -> None:
    _libmod.check(_libmod.load().ucf_comm_destroy(comm))


def build_id() -> str:
    return (_libmod.load().ucf_build_id() or b"").decode()


def drawdown_grid_multi(plans, tD, sv, rD, zD, zLay, with_stats: bool = False):
    """one sweep on len(plans) devices driven by this process (ucf_drawdown_grid_multi): h, dh [nt, nr, nz]"""
    lib = _libmod.load()
    tD, sv, rD, zD, zLay = _f64(tD), _i32(sv), _f64(rD), _f64(zD), _i32(zLay)
    nt, nr, nz = len(tD), len(rD), len(zD)
    arr = (C.c_void_p * len(plans))(*[p._h for p in plans])
    h = np.zeros((nt, nr, nz))
    dh = np.zeros((nt, nr, nz))
    st = UcfStats()
    _libmod.check(lib.ucf_drawdown_grid_multi(arr, len(plans), nt, tD, sv, nr, rD, nz, zD, zLay, h, dh, C.byref(st) if with_stats else None))
    if with_stats:
        return h, dh, {k: getattr(st, k) for k, _ in UcfStats._fields_}
    return h, dh


def drawdown_batch_multi(plans, tD, rD, sv, zD, zLay, with_stats: bool = False):
    """one point list on len(plans) devices driven by this process (ucf_drawdown_batch_multi): h, dh [npts, nz]"""
    lib = _libmod.load()
    tD, rD, sv, zD, zLay = _f64(tD), _f64(rD), _i32(sv), _f64(zD), _i32(zLay)
    npts, nz = len(tD), len(zD)
    arr = (C.c_void_p * len(plans))(*[p._h for p in plans])
    h = np.zeros((npts, nz))
    dh = np.zeros((npts, nz))
    st = UcfStats()
    _libmod.check(lib.ucf_drawdown_batch_multi(arr, len(plans), npts, tD, rD, sv, nz, zD, zLay, h, dh, C.byref(st) if with_stats else None))
    if with_stats:
        return h, dh, {k: getattr(st, k) for k, _ in UcfStats._fields_}
    return h, dh


def drawdown_multi(plans, t, r, z, dimensionless: bool = False):
    """the same dimensional observation points under many parameter sets (inversion / fitting):
    h, dh of shape [nplans, npts, nz]"""
    lib = _libmod.load()
    t, r, z = _f64(t), _f64(r), _f64(z)
    n, nz, npl = len(t), len(z), len(plans)
    if len(r) != n:
        raise ValueError("t and r must have equal length")
    arr = (C.c_void_p * npl)(*[p._h for p in plans])
    h = np.zeros((npl, n, nz))
    dh = np.zeros((npl, n, nz))
    _libmod.check(lib.ucf_drawdown_multi(arr, npl, n, t, r, nz, z, 1 if dimensionless else 0, h, dh))
    return h, dh


def debug_wynn(series, mode: str = "faithful"):
    """wynn_epsilon as finish_kernel runs it (epsilon table in registers, at most 12 terms): acc [n, 2], status [n]"""
    series = _f64(series)
    n, nterms = series.shape[0], series.shape[1]
    acc = np.zeros((n, 2)); st = np.zeros(n, np.int32)
    _libmod.check(_libmod.load().ucf_debug_wynn(1 if mode == "fast" else 0, n, nterms, series, acc, st))
    return acc, st


def debug_dehoog_tiles(M: int, alpha: float, tol: float, t, fp, mode: str = "faithful"):
    """deHoog_invlap as every grid call runs it (dehoog_tiles_kernel): fp [n, 2M+1, 2], T = 2 t; returns h [n], dh [n]"""
    t, fp = _f64(np.atleast_1d(t)), _f64(fp)
    n = len(t)
    h = np.zeros(n); dh = np.zeros(n)
    _libmod.check(_libmod.load().ucf_debug_dehoog_tiles(1 if mode == "fast" else 0, n, int(M), float(alpha), float(tol), t, fp, h, dh))
    return h, dh


def dehoog(M: int, alpha: float, tol: float, t, tee, fp) -> np.ndarray:
    lib = _libmod.load()
    t, tee, fp = _f64(np.atleast_1d(t)), _f64(np.atleast_1d(tee)), _f64(fp)
    n = len(t)
    out = np.zeros(n)
    _libmod.check(lib.ucf_dehoog(n, M, alpha, tol, t, tee, fp.reshape(n, 2 * M + 1, 2), out))
    return out


def wynn_epsilon(series):
    """series[n, nterms, 2] -> (acc[n,2], status[n])"""
    lib = _libmod.load()
    s = _f64(series)
    n, nt = s.shape[0], s.shape[1]
    acc = np.zeros((n, 2))
    st = np.zeros(n, np.int32)
    _libmod.check(lib.ucf_wynn_epsilon(n, nt, s, acc, st))
    return acc, st


def extraptozero(x, y):
    """x[R], y[n, R, 2] -> out[n, 2]"""
    lib = _libmod.load()
    x, y = _f64(x), _f64(y)
    n, R = y.shape[0], y.shape[1]
    out = np.zeros((n, 2))
    _libmod.check(lib.ucf_extraptozero(n, R, x, y, out))
    return out


def bessel_k01(z):
    """z[n,2] -> (K[n,2,2] = (K0,K1), ierr[n]); Amos cbesk(fnu=0, n=2, kode=1) on the device"""
    lib = _libmod.load()
    z = _f64(z)
    n = z.shape[0]
    k = np.zeros((n, 2, 2))
    ierr = np.zeros(n, np.int32)
    _libmod.check(lib.ucf_bessel_k01(n, z, k, ierr))
    return k, ierr


def fp64_fma_peak() -> float:
    lib = _libmod.load()
    v = C.c_double(0.0)
    _libmod.check(lib.ucf_fp64_fma_peak(C.byref(v)))
    return v.value


def logspace(lo: int, hi: int, n: int) -> np.ndarray:
    out = np.zeros(n)
    _libmod.check(_libmod.load().ucf_logspace(lo, hi, n, out))
    return out


def linspace(lo: float, hi: float, n: int) -> np.ndarray:
    out = np.zeros(n)
    _libmod.check(_libmod.load().ucf_linspace(lo, hi, n, out))
    return out


# ----------------------------------------------------------------------------- program Driver
class DeckResult:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def grids_from_deck(dk: Deck, deck_path: Optional[str] = None, ts: Optional[TimeSpec] = None,
                    sp: Optional[SpaceSpec] = None):
    """times, radii, depths as read_input builds them (driver_io.f90:385-523)"""
    dk.check_observation([], [], [])                 # what can be said before any grid exists
    if dk.timeseries:
        if ts is None:
            ts = TimeSpec.read(resolve(deck_path or ".", dk.timeFileName))
        t = logspace(ts.min_log, ts.max_log, ts.n) if ts.compute else _f64(ts.times)
        r = np.array([dk.rval])
        z = linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd)
    else:
        if sp is None:
            sp = SpaceSpec.read(resolve(deck_path or ".", dk.spaceFileName))
        t = np.array([dk.tval])
        if sp.compute:
            r = linspace(sp.min_r, sp.max_r, sp.n_r)
            z = linspace(sp.min_z, sp.max_z, sp.n_z)
        else:
            r, z = _f64(sp.r), _f64(sp.z)
    dk.check_observation(t, r, z, space_computed=(dk.timeseries or sp.compute))
    return t, r, z


def run_deck(deck_path: str, mode: str = "faithful") -> DeckResult:
    """what `./unconfined deck` computes: rows of (t, h, dh) for a time series or
    (z, r, h, dh) for a contour map, dimensional unless the deck says dimensionless.
    Contour mode uses each radius' own tanh-sinh interval (SURVEY.md quirk Q1 fixed)."""
    dk = Deck.read(deck_path)
    plan = Plan.from_deck(dk, mode)
    D = plan.derived
    t, r, z = grids_from_deck(dk, deck_path)
    tD, rD, zD = t / D.Tc, r / D.Lc, z / D.Lc
    zl = plan.zlay(zD)
    sv_t = plan.split_vector(tD)
    h, dh = plan.drawdown_grid(tD, sv_t, rD, zD, zl)
    h = h.reshape(len(tD) * len(rD), len(zD))
    dh = dh.reshape(len(tD) * len(rD), len(zD))
    sc = 1.0 if dk.dimless else D.Hc
    if dk.timeseries:
        hobs = screen_average_np(h, dk) * sc
        dobs = screen_average_np(dh, dk) * sc
        return DeckResult(deck=dk, plan=plan, t=(tD if dk.dimless else t), h=hobs, dh=dobs, raw_h=h, raw_dh=dh,
                          r_dim=float(r[0]), z_dim=z)
    nt, nr, nz = len(tD), len(rD), len(zD)
    return DeckResult(deck=dk, plan=plan, z=(zD if dk.dimless else z), r=(rD if dk.dimless else r),
                      h=(h * sc).reshape(nt, nr, nz), dh=(dh * sc).reshape(nt, nr, nz),
                      r_dim_all=r, z_dim=z, t_dim=float(t[0]))
