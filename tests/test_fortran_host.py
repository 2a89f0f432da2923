"""The thin Fortran host (unconfined_amd/fortran): ISO_C_BINDING over the C ABI, replacing the
reference's OpenMP loop nest by one call, writing the reference's own file format."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from golden_util import DECKS, e2e_gate_bounds, load_deck, load_e2e, rel_err
from unconfined_amd import output
from unconfined_amd.deck import Deck, SpaceSpec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "unconfined_amd", "fortran", "build", "ucf_host")
REF = os.path.join(ROOT, "oracle", "_ref", "O2", "unconfined")


def _have_host():
    if not os.path.exists(HOST) and os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "unconfined_amd", "fortran")], check=True, capture_output=True)
    return os.path.exists(HOST)


def _run(tmp_path, name, mode="faithful", extra=(), env=None):
    dk = Deck.read(os.path.join(DECKS, f"{name}.in"))
    for fn in (f"{name}.in", dk.timeFileName if dk.timeseries else dk.spaceFileName):
        shutil.copy(os.path.join(DECKS, fn), tmp_path)
    return subprocess.run([HOST, f"{name}.in", mode, *extra], cwd=tmp_path, capture_output=True, text=True, env=env), dk


def test_fortran_host_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    if not _have_host():
        pytest.skip("flang not available")
    res, dk = _run(tmp_path, "neuman74_partpen")
    assert res.returncode != 0
    assert "no CPU fallback" in res.stdout + res.stderr


@pytest.mark.parametrize("name", ["neuman74_partpen", "c3_moench", "mishra_fd30", "hantush_lay2", "theis_pulse", "mishra_malama",
                                  "hstorage_partpen_lay2", "c1_theis", "malama_partpen_b1", "contour_neuman"])
def test_fortran_host_header_is_the_reference_header(tmp_path, oracle, name):
    """the '#' parameter echo written by ucf_output.f90 equals unconfined_amd/output.py's line for line -- and that one is
    pinned byte for byte to files written by the reference binary (tests/test_output_format.py); where the reference
    binary travels with the repo its own header is compared too.  No GPU needed ("header" mode)."""
    if not _have_host():
        pytest.skip("flang not available")
    res, dk = _run(tmp_path, name, mode="header")
    assert res.returncode == 0, res.stdout + res.stderr
    mine = [ln for ln in open(tmp_path / dk.outFileName).read().split("\n") if ln]
    from unconfined_amd.abi import params_from_deck
    D = oracle.nondim(params_from_deck(dk))
    if dk.timeseries:
        _, ts, _ = load_deck(name)
        z = oracle.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd)
        want = output.timeseries_header(dk, D, dk.rval, dk.rval / D.Lc, z[0], z[0] / D.Lc, ts.n)
    else:
        sp = SpaceSpec.read(os.path.join(DECKS, dk.spaceFileName))
        want = output.contour_header(dk, D, oracle.linspace(sp.min_r, sp.max_r, sp.n_r), oracle.linspace(sp.min_z, sp.max_z, sp.n_z),
                                     dk.tval, dk.tval / D.Tc)
    assert mine == want
    if os.path.exists(REF):
        os.rename(tmp_path / dk.outFileName, tmp_path / "ours.out")
        subprocess.run([REF, f"{name}.in"], cwd=tmp_path, env=dict(os.environ, OMP_NUM_THREADS="4"), check=True, capture_output=True)
        ref = [ln for ln in open(tmp_path / dk.outFileName, errors="replace").read().split("\n") if ln.startswith("#")]
        assert mine == ref


@pytest.mark.gpu
@pytest.mark.parametrize("name,ngpu", [("neuman74_partpen", 1), ("c3_moench", 3), ("c1_theis", 2)])
def test_fortran_host_file_matches_reference(tmp_path, oracle, name, ngpu):
    """same deck, same file as ./unconfined writes: header bytes, time column, values within the end-to-end gate of
    tests/test_gpu_parity.py; with several plans (rehearsal: all on the one GPU) the rows of the time loop are sharded
    by ucf_drawdown_grid_multi and the file is the same"""
    assert _have_host(), "the Fortran host must have been built by __graft_entry__.build()"
    env = dict(os.environ, UCF_HOST_ONE_DEVICE="1")
    res, dk = _run(tmp_path, name, extra=(str(ngpu),), env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert f"ucf_host: {ngpu} GPU(s)" in res.stdout
    lines = [ln for ln in open(tmp_path / dk.outFileName).read().split("\n") if ln]
    rows = np.array([[float(x) for x in ln.split()[:3]] for ln in lines if not ln.startswith("#")])
    e2e = load_e2e(name)
    ir = int(np.argmin(np.abs(e2e["radii"] - dk.rval)))          # the fixture row of the deck's own radius
    assert e2e["radii"][ir] == dk.rval
    ref, bh, bd = e2e_gate_bounds(oracle, name, ir)
    assert rows.shape == ref.shape
    assert np.array_equal(rows[:, 0], ref[:, 0])                      # the time column is printed identically
    assert (rel_err(rows[:, 1], ref[:, 1], 1e-3) <= bh).all()
    assert (rel_err(rows[:, 2], ref[:, 2], 1e-3) <= bd).all()
    if ngpu > 1:                                                       # ... and equal to the single-plan file, bit for bit
        os.rename(tmp_path / dk.outFileName, tmp_path / "multi.out")
        res1, _ = _run(tmp_path, name, extra=("1",), env=env)
        assert res1.returncode == 0
        assert open(tmp_path / dk.outFileName).read() == open(tmp_path / "multi.out").read()
