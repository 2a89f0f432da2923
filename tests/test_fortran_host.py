"""The thin Fortran host (unconfined_amd/fortran): ISO_C_BINDING over the C ABI, replacing the
reference's OpenMP loop nest by one call."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from golden_util import DECKS, load_deck, load_e2e, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "unconfined_amd", "fortran", "build", "ucf_host")


def _have_host():
    if not os.path.exists(HOST) and os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "unconfined_amd", "fortran")], check=True, capture_output=True)
    return os.path.exists(HOST)


def _run(tmp_path, name, mode="faithful"):
    dk, ts, P = load_deck(name)
    for fn in (f"{name}.in", dk.timeFileName):
        shutil.copy(os.path.join(DECKS, fn), tmp_path)
    return subprocess.run([HOST, f"{name}.in", mode], cwd=tmp_path, capture_output=True, text=True), dk


def test_fortran_host_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    if not _have_host():
        pytest.skip("flang not available")
    res, dk = _run(tmp_path, "neuman74_partpen")
    assert res.returncode != 0
    assert "no CPU fallback" in res.stdout + res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["neuman74_partpen", "c3_moench", "c1_theis"])
def test_fortran_host_rows_match_reference(tmp_path, name):
    """same deck, same row format (ES14.07E2 / ES24.15E4) as ./unconfined; values within the
    end-to-end gate of tests/test_gpu_parity.py"""
    assert _have_host(), "the Fortran host must have been built by __graft_entry__.build()"
    res, dk = _run(tmp_path, name)
    assert res.returncode == 0, res.stdout + res.stderr
    rows = np.array([[float(x) for x in ln.split()[:3]] for ln in open(tmp_path / dk.outFileName) if not ln.startswith("#")])
    e2e = load_e2e(name)
    ir = int(np.argmin(np.abs(e2e["radii"] - dk.rval)))          # the fixture row of the deck's own radius
    assert e2e["radii"][ir] == dk.rval
    ref = e2e[f"O2_r{ir}"]
    assert rows.shape == ref.shape
    assert np.array_equal(rows[:, 0], ref[:, 0])                      # the time column is printed identically
    assert rel_err(rows[:, 1], ref[:, 1], 1e-3).max() < 1e-8
    assert rel_err(rows[:, 2], ref[:, 2], 1e-3).max() < 1e-6
