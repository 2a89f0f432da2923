"""ctypes mirrors of the POD structs in include/ucf.h (layout must match exactly;
tests/test_abi.py checks sizes/offsets against the compiled library)."""
from __future__ import annotations

import ctypes as C

UCF_MAX_MOENCH = 16
UCF_MAX_NZ = 32
UCF_MAX_LAP_M = 127
UCF_MAX_SCHEDULE = 100


class UcfParams(C.Structure):
    _fields_ = [
        ("model", C.c_int), ("MNtype", C.c_int), ("order", C.c_int), ("timeType", C.c_int),
        ("timePar", C.c_double * 2),
        ("Q", C.c_double), ("l", C.c_double), ("d", C.c_double), ("rw", C.c_double), ("rc", C.c_double),
        ("gammaSkin", C.c_double), ("b", C.c_double), ("Kr", C.c_double), ("kappa", C.c_double),
        ("Ss", C.c_double), ("Sy", C.c_double), ("beta", C.c_double),
        ("MoenchM", C.c_int), ("_pad0", C.c_int),
        ("MoenchAlpha", C.c_double * UCF_MAX_MOENCH),
        ("ac", C.c_double), ("ak", C.c_double), ("psia", C.c_double), ("psik", C.c_double), ("usL", C.c_double),
        ("M", C.c_int), ("k", C.c_int), ("R", C.c_int), ("nacc", C.c_int), ("ord", C.c_int),
        ("j0s", C.c_int * 2), ("_pad1", C.c_int),
        ("alpha", C.c_double), ("tol", C.c_double),
        ("rwobs", C.c_double), ("sF", C.c_double),
        ("timeParExt", C.c_double * (2 * UCF_MAX_SCHEDULE + 1)),
    ]


class UcfDerived(C.Structure):
    _fields_ = [
        ("Lc", C.c_double), ("Tc", C.c_double), ("Hc", C.c_double),
        ("sigma", C.c_double), ("alphaD", C.c_double), ("betaD", C.c_double),
        ("lD", C.c_double), ("dD", C.c_double), ("bD", C.c_double), ("rDw", C.c_double), ("rDwobs", C.c_double),
        ("acD", C.c_double), ("akD", C.c_double), ("lambdaD", C.c_double), ("psiaD", C.c_double),
        ("psikD", C.c_double), ("usLD", C.c_double), ("b1", C.c_double), ("PsiD", C.c_double),
        ("MoenchGamma", C.c_double * UCF_MAX_MOENCH),
        ("l_eff", C.c_double), ("d_eff", C.c_double), ("ac_eff", C.c_double),
        ("np", C.c_int), ("N", C.c_int), ("nj0z", C.c_int), ("nabs", C.c_int),
    ]


class UcfStats(C.Structure):
    _fields_ = [(n, C.c_longlong) for n in
                ("nan_scrubbed", "zero_vectors", "wynn_truncated", "wynn_sentinel", "wynn_early_exit", "wynn_all_zero")]


def params_from_deck(dk) -> UcfParams:
    """Deck (unconfined_amd.deck.Deck) -> ucf_params; values only, no arithmetic."""
    P = UcfParams()
    P.model, P.MNtype, P.order, P.timeType = dk.model, dk.MNtype, dk.order, dk.timeType
    tp = list(dk.timePar) + [0.0, 0.0]
    P.timePar[0], P.timePar[1] = tp[0], tp[1]
    if dk.timeType < 0:
        for i, v in enumerate(dk.timePar[:2 * UCF_MAX_SCHEDULE + 1]):
            P.timeParExt[i] = v
    for n in ("Q", "l", "d", "rw", "rc", "gammaSkin", "b", "Kr", "kappa", "Ss", "Sy", "beta",
              "ac", "ak", "psia", "psik", "usL", "alpha", "tol", "rwobs", "sF"):
        setattr(P, n, float(getattr(dk, n)))
    P.MoenchM = dk.MoenchM
    for i, a in enumerate(dk.MoenchAlpha[:UCF_MAX_MOENCH]):
        P.MoenchAlpha[i] = a
    P.M, P.k, P.R, P.nacc, P.ord = dk.M, dk.k, dk.R, dk.nacc, dk.ord
    P.j0s[0], P.j0s[1] = dk.j0s
    return P
