"""provenance of numerical constants baked into the HIP sources (no GPU needed)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exp_polynomial_matches_its_generator():
    """the coefficients of exp_pos() in ucf_fastpath.h are exactly what tools/gen_exp_poly.py prints, and the
    generator's own check says the rounded polynomial is good to < 2e-17 relative"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_exp_poly.py")], check=True, capture_output=True, text=True).stdout
    err = float(re.search(r"max rel err ([0-9.e+-]+)", out).group(1))
    assert err < 2e-17
    gen = [float(m) for m in re.findall(r"^s\d+ = ([0-9.e+-]+)$", out, flags=re.M)]
    assert len(gen) == 10
    src = open(os.path.join(ROOT, "unconfined_amd", "csrc", "ucf_fastpath.h")).read()
    body = src[src.index("UCF_DEV double exp_pos(double x)"):src.index("UCF_DEV fprim prim(double x, double y)")]
    lits = [float(m) for m in re.findall(r"K\(([0-9.e+-]+)\)", body)]
    assert lits == gen[::-1]                                   # s9 ... s0 in Horner order


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
