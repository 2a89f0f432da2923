"""provenance of numerical constants baked into the HIP sources (no GPU needed)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cody_waite_constants():
    """pi/2 = pio2_1 + pio2_1t to ~86 bits with a 33-bit head (fdlibm's split), as sincos_medium_ assumes"""
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    src = open(os.path.join(ROOT, "unconfined_amd", "csrc", "ucf_math.h")).read()
    body = src[src.index("UCF_DEV void sincos_medium_("):src.index("UCF_DEV void sincos_(double x, double* sn, double* cs)")]
    head = float(re.search(r"__builtin_fma\(-fn, ([0-9.e+-]+), x\)", body).group(1))
    tail = float(re.search(r"mulk\(fn, K\(([0-9.e+-]+)\)\)", body).group(1))
    half_pi = Decimal("1.57079632679489661923132169163975144209858469968755291")
    assert abs(Decimal(head) + Decimal(tail) - half_pi) < Decimal("1e-26")
    m = head.hex()                       # 33 significant bits: the low 20 bits of the 52-bit mantissa are zero
    assert int(m.split(".")[1].split("p")[0], 16) & ((1 << 20) - 1) == 0


def test_sincos_table_reduction_constants_and_entries():
    """sincos_tab_ (ucf_math.h): h = pi/128 = H1 + H2 with H1 = pio2_1 / 64 (a 33-bit head, so that x - k H1 is exact in one
    fma), k = rint(x * 128/pi); the 256-entry table the plan uploads holds the correctly rounded (sin, cos)(k h) with exact
    symmetry (zeros and ones exact: relative accuracy next to the zeros of sin and cos)"""
    from decimal import Decimal, getcontext
    import ctypes as C
    import numpy as np
    getcontext().prec = 60
    src = open(os.path.join(ROOT, "unconfined_amd", "csrc", "ucf_math.h")).read()
    body = src[src.index("UCF_DEV void sincos_tab_("):src.index("UCF_DEV void sincos_(double x, double* sn, double* cs)")]
    h1 = float(re.search(r"__builtin_fma\(-fn, ([0-9.e+-]+), x\)", body).group(1))
    h2 = float(re.search(r"__builtin_fma\(-fn, ([0-9.e+-]+), r\)", body).group(1))
    inv = float(re.search(r"UCF_SC_MAGIC_ADD\(t, fn, x, K\(([0-9.e+-]+)\)", body).group(1))
    pi = Decimal("3.14159265358979323846264338327950288419716939937510582")
    assert abs(Decimal(h1) + Decimal(h2) - pi / 128) < Decimal("1e-28")
    assert h1 == 1.57079632673412561417e+00 / 64 and h2 == 6.07710050650619224932e-11 / 64
    assert int(h1.hex().split(".")[1].split("p")[0], 16) & ((1 << 20) - 1) == 0        # 33 significant bits
    assert abs(Decimal(inv) - 128 / pi) < Decimal("1e-14")
    from unconfined_amd import lib
    so = lib.load()
    tab = np.zeros((256, 2))
    assert so.ucf_sincos_table(tab.ctypes.data_as(C.POINTER(C.c_double))) == 0
    k = np.arange(256)
    ang = k.astype(np.longdouble) * (np.longdouble("3.14159265358979323846264338327950288") / 128)
    assert np.array_equal(tab[:, 0], np.sin(ang).astype(np.float64) * (k % 128 != 0))     # sin(k pi) exactly 0
    want_c = np.cos(ang).astype(np.float64) * (k % 128 != 64)
    assert np.array_equal(tab[:, 1], want_c)
    assert np.array_equal(tab[:128, 0], -tab[128:, 0]) and np.array_equal(tab[:128, 1], -tab[128:, 1])   # half-turn
    assert np.array_equal(tab[64:128, 0], tab[:64, 1]) and np.array_equal(tab[64:128, 1], -tab[:64, 0])    # quarter-turn
    assert np.array_equal(tab[1:64, 0], tab[63:0:-1, 1])                                                   # sin(t) = cos(pi/2 - t)


def test_exp_table_reduction_constants():
    """exp_tab_ (ucf_math.h): x = k ln2/128 + r with ln2/128 = L1 + L2, L1 of 32 significant bits (k L1 exact for |k| < 2^21);
    the 128-entry table holds 2^(j/128) as (hi, lo), hi correctly rounded; the Taylor remainder r^6/720 at |r| = ln2/256 is < 1e-18"""
    from decimal import Decimal, getcontext
    import ctypes as C
    import numpy as np
    getcontext().prec = 60
    src = open(os.path.join(ROOT, "unconfined_amd", "csrc", "ucf_math.h")).read()
    body = src[src.index("UCF_DEV double exp_tab_("):src.index("UCF_DEV void sincos_(double x, double* sn, double* cs)")]
    l1 = float.fromhex(re.search(r"__builtin_fma\(-fn, (0x[0-9a-fp.+-]+), x\)", body).group(1))
    l2 = float(re.search(r"__builtin_fma\(-fn, ([0-9.e+-]+), r\)", body).group(1))
    inv = float(re.search(r"UCF_SC_MAGIC_ADD\(t, fn, x, K\(([0-9.e+-]+)\)", body).group(1))
    ln2 = Decimal(2).ln()
    assert abs(Decimal(l1) + Decimal(l2) - ln2 / 128) < Decimal("1e-28")
    assert int(l1.hex().split(".")[1].split("p")[0], 16) & ((1 << 21) - 1) == 0          # 32 significant bits
    assert abs(Decimal(inv) - 128 / ln2) < Decimal("1e-13")
    assert (float(ln2) / 256) ** 6 / 720 < 1e-18
    from unconfined_amd import lib
    so = lib.load()
    tab = np.zeros((128, 2))
    assert so.ucf_exp2_table(tab.ctypes.data_as(C.POINTER(C.c_double))) == 0
    for j in range(128):
        want = (ln2 * j / 128).exp()
        hi, lo = Decimal(float(tab[j, 0])), Decimal(float(tab[j, 1]))
        assert abs(hi - want) <= Decimal(float(np.spacing(tab[j, 0]))) / 2 * Decimal("1.0000001"), j   # hi correctly rounded
        assert abs(hi + lo - want) <= want * Decimal(2) ** -63, j                                       # hi + lo: 63 bits
    assert tab[0, 0] == 1.0 and tab[0, 1] == 0.0 and tab[64, 0] == 2.0 ** 0.5

