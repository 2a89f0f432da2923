"""CPU-side checks of the boundary: the shared library loads, exports every symbol that
include/ucf.h declares, the ctypes struct mirrors match the C layout, and compute entry
points fail loudly (no fallback) when there is no GPU."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from unconfined_amd import lib as ucflib
from unconfined_amd.abi import UcfDerived, UcfParams, UcfStats, params_from_deck
from unconfined_amd.deck import Deck

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    if not os.path.exists(ucflib.LIB_PATH):
        ucflib.build()
    return ucflib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ucf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ucf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(so):
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(so, s), f"libucf.so does not export {s}"
    assert sorted(ucflib.EXPORTS) == syms
    assert so.ucf_version() == 100


def test_struct_layouts_match_header():
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "ucf.h"
    int main(void){
      printf("%zu %zu %zu\n", sizeof(ucf_params), sizeof(ucf_derived), sizeof(ucf_stats));
      printf("%zu %zu %zu %zu %zu %zu\n", offsetof(ucf_params,timePar), offsetof(ucf_params,MoenchAlpha),
             offsetof(ucf_params,ac), offsetof(ucf_params,M), offsetof(ucf_params,alpha), offsetof(ucf_params,sF));
      printf("%zu %zu %zu\n", offsetof(ucf_derived,MoenchGamma), offsetof(ucf_derived,l_eff), offsetof(ucf_derived,np));
      return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.c")
        open(p, "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), p, "-o", os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    vals = [int(x) for x in out]
    assert vals[:3] == [C.sizeof(UcfParams), C.sizeof(UcfDerived), C.sizeof(UcfStats)]
    assert vals[3:9] == [UcfParams.timePar.offset, UcfParams.MoenchAlpha.offset, UcfParams.ac.offset,
                         UcfParams.M.offset, UcfParams.alpha.offset, UcfParams.sF.offset]
    assert vals[9:] == [UcfDerived.MoenchGamma.offset, UcfDerived.l_eff.offset, UcfDerived.np.offset]


def test_input_validation_mirrors_reference_stops(so):
    """driver_io.f90:88-333: each `stop` of read_input is a distinct negative status here"""
    base = Deck.read(os.path.join(ROOT, "tests", "golden", "decks", "neuman74_partpen.in"))
    cases = [
        (dict(model=7), -1), (dict(l=1.0, d=2.0), -2), (dict(b=-1.0), -3), (dict(kappa=0.0), -3),
        (dict(model=6, MNtype=2, order=2), -4), (dict(beta=-1.0), -5), (dict(model=3, MoenchM=0), -6),
        (dict(M=1), -7), (dict(k=4, R=4), -8), (dict(R=0), -8), (dict(nacc=0), -9),
        (dict(model=6, MNtype=0), -10), (dict(M=128), -10), (dict(timeType=-201), -10), (dict(timeType=0), -10),
        (dict(timeType=-102, timePar=[1.0, 1.0, 3.0, 1.0, 2.0]), -11),
    ]
    for kw, want in cases:
        P = params_from_deck(base.replace(**kw))
        h = C.c_void_p()
        rc = so.ucf_plan_create(C.byref(P), C.byref(h))
        assert rc == want, (kw, rc, so.ucf_last_error())
        assert not h.value
        assert so.ucf_last_error()


def test_no_cpu_fallback(so):
    """without a usable HIP device the product must refuse, not compute on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    P = params_from_deck(Deck.read(os.path.join(ROOT, "tests", "golden", "decks", "neuman74_partpen.in")))
    h = C.c_void_p()
    rc = so.ucf_plan_create(C.byref(P), C.byref(h))
    assert rc == -12 and b"no CPU fallback" in so.ucf_last_error()
    v = C.c_double()
    assert so.ucf_fp64_fma_peak(C.byref(v)) == -12


def test_product_does_not_touch_the_oracle():
    """the product package must not import / link / execute anything under oracle/"""
    pkg = os.path.join(ROOT, "unconfined_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".f90")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle_lib" not in text and "ucf_oracle" not in text and "libucf_oracle" not in text, fn
    out = subprocess.run(["ldd", ucflib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_host_only_entries_work_without_a_gpu(so, oracle):
    """what the reference computes in read_input needs no device: the partition rule of the multi-GPU entry points,
    read_input's checks + non-dimensionalisation (bit for bit the oracle's, hence the reference's), and the header that a
    reference-side Fortran binding would include compiles as plain C"""
    lo, hi = C.c_int(), C.c_int()
    assert so.ucf_shard_rows(1024, 8, 3, C.byref(lo), C.byref(hi)) == 0 and (lo.value, hi.value) == (384, 512)
    assert so.ucf_shard_rows(10, 4, 3, C.byref(lo), C.byref(hi)) == 0 and (lo.value, hi.value) == (9, 10)
    assert so.ucf_shard_rows(2, 4, 3, C.byref(lo), C.byref(hi)) == 0 and lo.value == hi.value        # an empty shard
    for bad in ((10, 0, 0), (10, 2, 2), (-1, 2, 0)):
        assert so.ucf_shard_rows(*bad, C.byref(lo), C.byref(hi)) == -11
    for name in ("neuman74_partpen", "c3_moench", "mishra_malama", "hstorage_partpen_lay1"):
        P = params_from_deck(Deck.read(os.path.join(ROOT, "tests", "golden", "decks", name + ".in")))
        D = UcfDerived()
        assert so.ucf_nondimensionalise(C.byref(P), C.byref(D)) == 0
        Do = oracle.nondim(P)
        for f, _ in UcfDerived._fields_:
            a, b = getattr(D, f), getattr(Do, f)
            assert (list(a) == list(b)) if f == "MoenchGamma" else (a == b), (name, f)
    P.b = -1.0
    assert so.ucf_nondimensionalise(C.byref(P), C.byref(D)) == -3
    assert len(so.ucf_build_id()) == 16


def test_device_entries_say_no_device_without_a_gpu(so):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    n = C.c_int(7)
    assert so.ucf_device_count(C.byref(n)) == -12 and n.value == 0
    P = params_from_deck(Deck.read(os.path.join(ROOT, "tests", "golden", "decks", "neuman74_partpen.in")))
    h = C.c_void_p()
    assert so.ucf_plan_create_on(C.byref(P), 0, C.byref(h)) == -12 and not h.value
    assert b"no CPU fallback" in so.ucf_last_error()


def test_multi_device_entries_check_their_arguments(so):
    """ucf_drawdown_grid_multi / ucf_drawdown_batch_multi: argument errors are reported before any device is touched"""
    import numpy as np
    d = np.zeros(4)
    i = np.ones(4, np.int32)
    dp = ip = lambda a: a                  # (lib.py declares these arguments as numpy arrays)
    none = C.POINTER(C.c_void_p)()
    assert so.ucf_drawdown_batch_multi(none, 2, 4, dp(d), dp(d), ip(i), 1, dp(d), ip(i), dp(d), dp(d), None) == -11
    assert so.ucf_drawdown_grid_multi(none, 2, 2, dp(d), ip(i), 2, dp(d), 1, dp(d), ip(i), dp(d), dp(d), None) == -11
    plans = (C.c_void_p * 2)(None, None)
    assert so.ucf_drawdown_batch_multi(plans, 2, 4, dp(d), dp(d), ip(i), 1, dp(d), ip(i), dp(d), dp(d), None) == -11
    assert b"plans[0]" in so.ucf_last_error()
    assert so.ucf_drawdown_batch_multi(plans, 0, 4, dp(d), dp(d), ip(i), 1, dp(d), ip(i), dp(d), dp(d), None) == -11

