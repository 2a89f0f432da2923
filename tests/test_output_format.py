"""Byte-level check of unconfined_amd/output.py against files written by the reference binary
(oracle/_ref/O2/unconfined, built by `make -C oracle ref`; the binary travels with the repo, its
sources do not).  Numbers are taken from the reference file itself, so this tests formatting only."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from golden_util import DECKS, load_deck
from unconfined_amd import output

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "O2", "unconfined")


@pytest.mark.parametrize("name", ["neuman74_partpen", "c3_moench", "mishra_fd30", "hantush_lay2", "theis_pulse",
                                  "mishra_malama", "hstorage_partpen_lay2"])
def test_timeseries_file_matches_reference_bytes(tmp_path, oracle, name):
    if not os.path.exists(REF):
        pytest.skip("reference binary not built (oracle/_ref)")
    dk, ts, P = load_deck(name)
    for fn in (f"{name}.in", dk.timeFileName):
        shutil.copy(os.path.join(DECKS, fn), tmp_path)
    env = dict(os.environ, OMP_NUM_THREADS="4")
    subprocess.run([REF, f"{name}.in"], cwd=tmp_path, env=env, check=True, capture_output=True)
    ref_lines = open(tmp_path / dk.outFileName, errors="replace").read().split("\n")
    rows_ref = [ln for ln in ref_lines if ln and not ln.startswith("#")]
    vals = np.array([[float(x) for x in ln.split()] for ln in rows_ref])
    D = oracle.nondim(P)
    z = oracle.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd)
    head = output.timeseries_header(dk, D, dk.rval, dk.rval / D.Lc, z[0], z[0] / D.Lc, len(vals))
    mine = head + output.timeseries_rows(vals[:, 0], vals[:, 1], vals[:, 2])
    ref_clean = [ln for ln in ref_lines if ln != ""]
    assert len(mine) == len(ref_clean)
    for a, b in zip(mine, ref_clean):
        assert a == b, (a, b)


def test_es_edit_descriptors():
    assert output.rfmt(0.1) == " 1.0000000E-01"
    assert output.rfmt(-123456.789) == "-1.2345679E+05"
    assert output.hfmt(4.417735711365508e-03) == " 4.417735711365508E-0003"
    assert output.hfmt(-1.436398020545476) == "-1.436398020545476E+0000"
    assert output.rfmt(0.0) == " 0.0000000E+00"
    assert output.rfmt(1e-100) == "*" * 14          # exponent does not fit E2
    assert output.hfmt(float("nan")).strip() == "NaN"


def test_contour_file_matches_reference_bytes(tmp_path, oracle):
    """contour mode (timeseries = F): header of driver_io.f90:760-845 and rows of driver.f90:259-271.
    (Values of the reference are only trustworthy for its first radius, SURVEY.md quirk Q1 -- this test
    re-formats the reference's own numbers.)"""
    if not os.path.exists(REF):
        pytest.skip("reference binary not built (oracle/_ref)")
    from unconfined_amd.deck import Deck, SpaceSpec
    name = "contour_neuman"
    dk = Deck.read(os.path.join(DECKS, f"{name}.in"))
    for fn in (f"{name}.in", dk.spaceFileName):
        shutil.copy(os.path.join(DECKS, fn), tmp_path)
    subprocess.run([REF, f"{name}.in"], cwd=tmp_path, env=dict(os.environ, OMP_NUM_THREADS="4"), check=True, capture_output=True)
    ref_lines = [ln for ln in open(tmp_path / dk.outFileName, errors="replace").read().split("\n") if ln != ""]
    vals = np.array([[float(x) for x in ln.split()] for ln in ref_lines if not ln.startswith("#")])
    from unconfined_amd.abi import params_from_deck
    D = oracle.nondim(params_from_deck(dk))
    sp = SpaceSpec.read(os.path.join(DECKS, dk.spaceFileName))
    r = oracle.linspace(sp.min_r, sp.max_r, sp.n_r)
    z = oracle.linspace(sp.min_z, sp.max_z, sp.n_z)
    head = output.contour_header(dk, D, r, z, dk.tval, dk.tval / D.Tc)
    h = vals[:, 2].reshape(len(r), len(z)); dh = vals[:, 3].reshape(len(r), len(z))
    mine = head + output.contour_rows(z, r, h, dh)
    assert len(mine) == len(ref_lines)
    for a, b in zip(mine, ref_lines):
        assert a == b, (a, b)
