"""Bad decks: where `./unconfined` prints ERROR and stops (reference driver_io.f90:355-394,443-449,494-519) the
Python host raises DeckError and the Fortran host exits non-zero with the message -- nobody computes numbers for an
observation point outside the aquifer or a radius inside the well."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from golden_util import DECKS
from unconfined_amd.deck import Deck, DeckError, SpaceSpec, TimeSpec
from unconfined_amd.engine import grids_from_deck

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "unconfined_amd", "fortran", "build", "ucf_host")

BAD_TS = [("zTop", dict(zTop=10.0, zBot=20.0), "at or above bottom"),
          ("above_b", dict(zTop=1.0e4), "between 0 and b"),
          ("below_0", dict(zBot=-1.0), "between 0 and b"),
          ("zOrd", dict(piezometer=False, zOrd=0), "quadrature points"),
          ("rwobs", dict(rwobs=0.0), "monitoring well radius"),
          ("sF", dict(sF=-1.0), "shape factor"),
          ("r_in_well", dict(rval=0.1), "r must be > rw")]


@pytest.mark.parametrize("tag,change,msg", BAD_TS)
def test_bad_time_series_decks(tmp_path, tag, change, msg):
    dk = Deck.read(os.path.join(DECKS, "neuman74_partpen.in")).replace(**change)
    ts = TimeSpec.read(os.path.join(DECKS, dk.timeFileName))
    with pytest.raises(DeckError) as e:
        grids_from_deck(dk, ts=ts)
    assert msg in str(e.value)
    if os.path.exists(HOST):
        dk.write(tmp_path / "bad.in")
        shutil.copy(os.path.join(DECKS, dk.timeFileName), tmp_path)
        res = subprocess.run([HOST, "bad.in", "header"], cwd=tmp_path, capture_output=True, text=True)
        assert res.returncode != 0 and msg in res.stdout, res.stdout


def test_bad_times_and_contour_grids(tmp_path):
    dk = Deck.read(os.path.join(DECKS, "neuman74_partpen.in"))
    with pytest.raises(DeckError):
        grids_from_deck(dk, ts=TimeSpec(False, -1, 1, 3, times=[1.0, -2.0, 3.0]))
    ck = Deck.read(os.path.join(DECKS, "contour_neuman.in"))
    sp = SpaceSpec.read(os.path.join(DECKS, ck.spaceFileName))
    grids_from_deck(ck, sp=sp)                                                       # the fixture itself is fine
    for bad in (dict(min_z=-1.0), dict(max_z=ck.b * 1.5)):
        with pytest.raises(DeckError):
            grids_from_deck(ck, sp=SpaceSpec(**{**sp.__dict__, **bad}))
    listed = SpaceSpec(False, 0, 1, 1, 0, 1, 1, r=[0.01, 5.0], z=[1.0, 2.0])
    with pytest.raises(DeckError) as e:
        grids_from_deck(ck, sp=listed)
    assert "rw" in str(e.value)
