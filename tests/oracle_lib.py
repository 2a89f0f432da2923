"""ctypes binding of the CPU oracle (oracle/libucf_oracle.so) -- TEST CODE ONLY.

The product package never imports this module; it is the checker.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from unconfined_amd.abi import UcfDerived, UcfParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class Stage(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_double)) for n in ("p", "fa", "tmp", "finint", "glarea", "infint", "totlap")]


def _build():
    subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, capture_output=True)


def load(quad: bool = False) -> C.CDLL:
    name = "libucf_oracle_q.so" if quad else "libucf_oracle.so"
    path = os.path.join(ORACLE_DIR, name)
    src = os.path.join(ORACLE_DIR, "ucf_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        _build()
    lib = C.CDLL(path)
    lib.ucfo_maxexp.restype = C.c_double
    lib.ucfo_nondim.argtypes = [C.POINTER(UcfParams), C.POINTER(UcfDerived)]
    lib.ucfo_zlay.argtypes = [C.POINTER(UcfDerived), C.c_int, _dp, _ip]
    lib.ucfo_j0_zeros.argtypes = [C.c_int, _dp]
    lib.ucfo_split_vector.argtypes = [C.c_int * 2, C.c_int, _dp, _ip]
    lib.ucfo_linspace.argtypes = [C.c_double, C.c_double, C.c_int, _dp]
    lib.ucfo_logspace.argtypes = [C.c_int, C.c_int, C.c_int, _dp]
    lib.ucfo_pvalues.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double, _dp]
    lib.ucfo_dehoog.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _dp]
    lib.ucfo_dehoog.restype = C.c_double
    lib.ucfo_tanh_sinh.argtypes = [C.c_int, C.c_double, _dp, C.c_void_p]
    lib.ucfo_gauss_lobatto.argtypes = [C.c_int, _dp, _dp]
    lib.ucfo_wynn_epsilon.argtypes = [C.c_int, _dp, _dp, C.POINTER(C.c_int)]
    lib.ucfo_extraptozero.argtypes = [C.c_int, _dp, _dp, _dp]
    lib.ucfo_lap_hank_soln.argtypes = [C.POINTER(UcfParams), C.POINTER(UcfDerived), C.c_double, C.c_double,
                                       C.c_int, _dp, C.c_int, _dp, _ip, _dp]
    lib.ucfo_point.argtypes = [C.POINTER(UcfParams), C.POINTER(UcfDerived), _dp, C.c_double, C.c_double, C.c_int,
                               C.c_int, _dp, _ip, _dp, _dp, C.POINTER(Stage)]
    lib.ucfo_batch.argtypes = [C.POINTER(UcfParams), C.c_int, _dp, _dp, _ip, C.c_int, _dp, _ip, _dp, _dp, C.c_int]
    return lib


class Oracle:
    """thin numpy-facing wrapper"""

    def __init__(self, quad: bool = False):
        self.lib = load(quad)

    def nondim(self, P: UcfParams) -> UcfDerived:
        D = UcfDerived()
        self.lib.ucfo_nondim(C.byref(P), C.byref(D))
        return D

    def zlay(self, D, zD):
        zD = np.ascontiguousarray(zD, np.float64)
        out = np.zeros(len(zD), np.int32)
        self.lib.ucfo_zlay(C.byref(D), len(zD), zD, out)
        return out

    def j0_zeros(self, n):
        out = np.zeros(n)
        self.lib.ucfo_j0_zeros(n, out)
        return out

    def split_vector(self, j0s, tD):
        tD = np.ascontiguousarray(tD, np.float64)
        out = np.zeros(len(tD), np.int32)
        self.lib.ucfo_split_vector((C.c_int * 2)(*j0s), len(tD), tD, out)
        return out

    def logspace(self, lo, hi, n):
        out = np.zeros(n)
        self.lib.ucfo_logspace(lo, hi, n, out)
        return out

    def linspace(self, lo, hi, n):
        out = np.zeros(n)
        self.lib.ucfo_linspace(lo, hi, n, out)
        return out

    def pvalues(self, tee, M, alpha, tol):
        out = np.zeros((2 * M + 1, 2))
        self.lib.ucfo_pvalues(tee, M, alpha, tol, out)
        return out

    def dehoog(self, M, alpha, tol, t, tee, fp):
        fp = np.ascontiguousarray(fp, np.float64)
        return self.lib.ucfo_dehoog(M, alpha, tol, t, tee, fp)

    def tanh_sinh(self, k, s, with_abscissae=True):
        n = 2 ** k - 1
        w = np.zeros(n)
        a = np.zeros(n)
        self.lib.ucfo_tanh_sinh(k, s, w, a.ctypes.data if with_abscissae else None)
        return w, a

    def gauss_lobatto(self, order):
        x = np.zeros(order - 2)
        w = np.zeros(order - 2)
        self.lib.ucfo_gauss_lobatto(order, x, w)
        return x, w

    def wynn(self, series):
        series = np.ascontiguousarray(series, np.float64)
        out = np.zeros(2)
        st = C.c_int(0)
        self.lib.ucfo_wynn_epsilon(len(series), series, out, C.byref(st))
        return out, st.value

    def extrap(self, x, y):
        x = np.ascontiguousarray(x, np.float64)
        y = np.ascontiguousarray(y, np.float64)
        out = np.zeros(2)
        self.lib.ucfo_extraptozero(len(x), x, y, out)
        return out

    def soln(self, P, D, a, rD, p, zD, zLay):
        p = np.ascontiguousarray(p, np.float64)
        zD = np.ascontiguousarray(zD, np.float64)
        zLay = np.ascontiguousarray(zLay, np.int32)
        np_ = p.shape[0]
        out = np.zeros((len(zD), np_, 2))
        rc = self.lib.ucfo_lap_hank_soln(C.byref(P), C.byref(D), a, rD, np_, p, len(zD), zD, zLay, out)
        if rc:
            raise RuntimeError(f"oracle: unsupported model (rc={rc})")
        return out

    def point(self, P, D, j0z, tD, rD, sv, zD, zLay, stages=False):
        zD = np.ascontiguousarray(zD, np.float64)
        zLay = np.ascontiguousarray(zLay, np.int32)
        j0z = np.ascontiguousarray(j0z, np.float64)
        nz = len(zD)
        h = np.zeros(nz)
        dh = np.zeros(nz)
        st = None
        bufs = {}
        if stages:
            np_, N, R, nacc = D.np, D.N, P.R, P.nacc
            shapes = dict(p=(np_, 2), fa=(N, nz, np_, 2), tmp=(R, nz, np_, 2), finint=(nz, np_, 2),
                          glarea=(nacc, nz, np_, 2), infint=(nz, np_, 2), totlap=(nz, np_, 2))
            st = Stage()
            for k, shp in shapes.items():
                bufs[k] = np.zeros(shp)
                setattr(st, k, bufs[k].ctypes.data_as(C.POINTER(C.c_double)))
        rc = self.lib.ucfo_point(C.byref(P), C.byref(D), j0z, tD, rD, int(sv), nz, zD, zLay, h, dh,
                                 C.byref(st) if st is not None else None)
        if rc:
            raise RuntimeError(f"oracle: unsupported model (rc={rc})")
        return (h, dh, bufs) if stages else (h, dh)

    STAT_NAMES = ("nan_scrubbed", "zero_vectors", "wynn_truncated", "wynn_sentinel", "wynn_early_exit", "wynn_all_zero")

    def batch_with_stats(self, P, tD, rD, sv, zD, zLay, threads=0):
        """batch + the counts of the in-band rules it took (same counters as ucf_stats)"""
        self.lib.ucfo_stats_reset()
        h, dh = self.batch(P, tD, rD, sv, zD, zLay, threads)
        out = (C.c_longlong * 6)()
        self.lib.ucfo_stats_get(out)
        return h, dh, dict(zip(self.STAT_NAMES, [int(v) for v in out]))

    def batch(self, P, tD, rD, sv, zD, zLay, threads=0):
        tD = np.ascontiguousarray(tD, np.float64)
        rD = np.ascontiguousarray(rD, np.float64)
        sv = np.ascontiguousarray(sv, np.int32)
        zD = np.ascontiguousarray(zD, np.float64)
        zLay = np.ascontiguousarray(zLay, np.int32)
        n, nz = len(tD), len(zD)
        h = np.zeros((n, nz))
        dh = np.zeros((n, nz))
        rc = self.lib.ucfo_batch(C.byref(P), n, tD, rD, sv, nz, zD, zLay, h, dh, threads)
        if rc:
            raise RuntimeError(f"oracle: unsupported model (rc={rc})")
        return h, dh
