"""The oracle (oracle/ucf_oracle.c) pinned against the reference itself.

Fixtures under tests/golden/ were produced by oracle/gen_golden.py from the
UNMODIFIED reference compiled with flang (oracle/Makefile `make ref`):
stage vectors through oracle/ref_harness.f90 (the reference's own read_input,
lap_hank_soln, deHoog_pvalues/invlap, tanh_sinh_setup, gauss_lobatto_setup,
wynn_epsilon, extraptozero) and end-to-end .out files of the reference binary.
Stage parity is required BIT FOR BIT; end-to-end parity to the 16 digits the
reference prints.
"""
import numpy as np
import pytest

from golden_util import bits_equal, deck_names, load_deck, load_e2e, load_stages, rel_err, ulps, unhx

NAMES = deck_names()


@pytest.mark.parametrize("name", NAMES)
def test_nondimensionalisation_j0zeros_splitvector(oracle, name):
    """driver_io.f90:531-567, 575-586, 628-647, 654-664"""
    meta, z = load_stages(name)
    dk, ts, P = load_deck(name)
    D = oracle.nondim(P)
    for key, hexval in meta["scalars_hex"].items():
        if hasattr(D, key):
            assert getattr(D, key) == unhx(hexval), key
    for i, g in enumerate(z["par_MoenchGamma"]) if dk.model == 3 else []:
        assert D.MoenchGamma[i] == g
    assert bits_equal(oracle.j0_zeros(D.nj0z), z["par_j0z"])
    # times: logspace (utility.f90:51-57) then /Tc
    t = oracle.logspace(ts.min_log, ts.max_log, ts.n)
    assert bits_equal(t, z["par_t"])
    assert bits_equal(t / D.Tc, z["par_tD"])
    assert np.array_equal(oracle.split_vector(list(dk.j0s), z["par_tD"]), z["par_sv"])
    # depths: linspace(zBot,zTop,zOrd)/Lc and the layer rule
    zz = oracle.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd)
    assert bits_equal(zz / D.Lc, z["par_zD"])
    assert np.array_equal(oracle.zlay(D, z["par_zD"]), z["par_zLay"])
    assert bits_equal(np.array([dk.rval / D.Lc]), z["par_rD"])


@pytest.mark.parametrize("name", NAMES)
def test_pvalues_and_samples_bit_exact(oracle, name):
    """invlap.f90:154-172 and laplace_hankel_solutions.f90:30-120 (+ time.f90:34-80)"""
    meta, z = load_stages(name)
    dk, ts, P = load_deck(name)
    D = oracle.nondim(P)
    for it, tee in enumerate(z["pv_tee"]):
        assert bits_equal(oracle.pvalues(tee, dk.M, dk.alpha, dk.tol), z["pv_p"][it])
    n_nan = 0
    for i, (tD, a, rD) in enumerate(zip(z["soln_tD"], z["soln_a"], z["soln_rD"])):
        p = oracle.pvalues(2 * tD, dk.M, dk.alpha, dk.tol)
        fp = oracle.soln(P, D, a, rD, p, z["par_zD"], z["par_zLay"])
        ref = z["soln_fp"][i]
        assert float(ulps(fp, ref).max()) == 0.0, (name, i, a, tD)
        n_nan += int(np.isnan(ref).sum())
    if dk.model in (1, 3, 5, 6) and not (dk.model == 6 and dk.MNtype == 1):
        assert n_nan > 0, "the overflow probe (a~3000) must exercise the Inf/NaN path"


@pytest.mark.parametrize("name", NAMES)
def test_quadrature_tables_bit_exact(oracle, name):
    """integration.f90:31-67 and 70-120"""
    meta, z = load_stages(name)
    dk, ts, P = load_deck(name)
    arg = float(z["ts_arg"][0])
    for j in range(1, dk.R + 1):
        w, a = oracle.tanh_sinh(dk.k - dk.R + j, arg)
        assert bits_equal(w, z[f"ts_w{j}"]) and bits_equal(a, z[f"ts_a{j}"])
        assert abs(w.sum() - 2.0) < 1e-14
    x, w = oracle.gauss_lobatto(dk.ord)
    assert bits_equal(x, z["gl_x"]) and bits_equal(w, z["gl_w"])


def test_wynn_extrap_dehoog_bit_exact(oracle):
    """integration.f90:125-189, 192-237; invlap.f90:46-152 -- incl. NaN truncation,
    the -999999.9 sentinel, the absolute-epsilon early exit, NaN scrub and zero vector"""
    import os
    from golden_util import GOLD
    z = np.load(os.path.join(GOLD, "stages_generic.npz"))
    nw, ne, nd = z["counts"]
    statuses = set()
    for i in range(nw):
        out, st = oracle.wynn(z[f"wynn_in_{i}"])
        statuses.add(st)
        assert bits_equal(out, z[f"wynn_out_{i}"]), ("wynn", i)
    assert statuses == {0, 1, 2, 3}, statuses
    for i in range(ne):
        assert bits_equal(oracle.extrap(z[f"extrap_x_{i}"], z[f"extrap_y_{i}"]), z[f"extrap_out_{i}"]), ("extrap", i)
    for i in range(nd):
        M, alpha, tol, t, tee = z[f"dehoog_par_{i}"]
        out = oracle.dehoog(int(M), alpha, tol, t, tee, z[f"dehoog_fp_{i}"])
        assert bits_equal(np.array([out]), z[f"dehoog_out_{i}"]), ("dehoog", i)


def test_cbesk_k0_k1_bit_exact(oracle):
    """Amos cbesk/cbknu for (fnu=0, n=2, kode=1), cbessel.f90:877,5036: power series for |z| <= 2 and
    Miller backward recurrence beyond, against the reference's routine on 105 right-half-plane points"""
    import ctypes as C
    import os
    from golden_util import GOLD
    z = np.load(os.path.join(GOLD, "stages_generic.npz"))
    lib = oracle.lib
    lib.ucfo_cbesk01.argtypes = [C.c_double, C.c_double, np.ctypeslib.ndpointer(np.float64)]
    nbad = 0
    for i, (zr, zi) in enumerate(z["cbesk_z"]):
        out = np.zeros(4)
        ierr = lib.ucfo_cbesk01(float(zr), float(zi), out)
        nzr, ierr_ref = z["cbesk_nz_ierr"][i]
        if ierr_ref != 0:
            assert ierr != 0
            continue
        assert ierr == 0, (zr, zi)
        if not bits_equal(out.reshape(2, 2), z["cbesk_k"][i]):
            nbad += 1
            assert np.max(np.abs(out.reshape(2, 2) - z["cbesk_k"][i]) / np.abs(z["cbesk_k"][i]).max()) < 4e-16, (zr, zi)
    assert nbad <= 3, nbad       # a handful of points differ in the last bit (libm pow/log paths), none by more


def _oracle_rows(oracle, name, ir, e2e):
    dk, ts, P = load_deck(name)
    P.l = dk.l  # noqa
    D = oracle.nondim(P)
    t = oracle.logspace(ts.min_log, ts.max_log, ts.n)
    tD = t / D.Tc
    sv = oracle.split_vector(list(dk.j0s), tD)
    zz = oracle.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd)
    zD = zz / D.Lc
    zl = oracle.zlay(D, zD)
    r = float(e2e["radii"][ir])
    rD = np.full_like(tD, r / D.Lc)
    return dk, D, t, tD, rD, sv, zD, zl


E2E_SMALL = [n for n in NAMES if not n.startswith(("c2_", "c3_", "c4_", "c5_", "malama_k10", "mishra_malama"))]


@pytest.mark.parametrize("name", E2E_SMALL)
def test_end_to_end_vs_reference_binary(oracle, name):
    """whole loop body, driver.f90:100-273, against the reference's .out (flang -O2).
    The .out carries 16 significant digits (HFMT = ES24.15E4)."""
    e2e = load_e2e(name)
    if e2e is None:
        pytest.skip("no e2e fixture")
    dk, D, t, tD, rD, sv, zD, zl = _oracle_rows(oracle, name, 0, e2e)
    _, _, P = load_deck(name)
    h, dh = oracle.batch(P, tD, rD, sv, zD, zl)
    from unconfined_amd.host import screen_average_np
    hobs = screen_average_np(h, dk) * (1.0 if dk.dimless else D.Hc)
    dobs = screen_average_np(dh, dk) * (1.0 if dk.dimless else D.Hc)
    ref = e2e["O2_r0"]
    assert ref.shape[0] == len(t)
    # printed with 16 significant digits -> agreement to ~1e-15 relative means the same double
    scale_h = np.maximum(np.abs(ref[:, 1]), 1e-300)
    scale_d = np.maximum(np.abs(ref[:, 2]), 1e-300)
    assert np.max(np.abs(hobs - ref[:, 1]) / scale_h) < 2e-15, name
    assert np.max(np.abs(dobs - ref[:, 2]) / scale_d) < 2e-15, name


@pytest.mark.parametrize("name,nsub", [("c2_neuman74_fullpen", 64), ("c3_moench", 32), ("c4_malama_partpen", 32),
                                       ("c5_mishra_fd64", 16), ("malama_k10", 8), ("mishra_malama", 8)])
def test_end_to_end_configs_subsampled(oracle, name, nsub):
    """the BASELINE.json configurations: every radius of the fixture, a strided subsample of times"""
    e2e = load_e2e(name)
    if e2e is None:
        pytest.skip("no e2e fixture")
    _, _, P = load_deck(name)
    from unconfined_amd.host import screen_average_np
    for ir in range(len(e2e["radii"])):
        dk, D, t, tD, rD, sv, zD, zl = _oracle_rows(oracle, name, ir, e2e)
        idx = np.unique(np.linspace(0, len(t) - 1, nsub).astype(int))
        h, dh = oracle.batch(P, tD[idx], rD[idx], sv[idx], zD, zl)
        sc = 1.0 if dk.dimless else D.Hc
        ref = e2e[f"O2_r{ir}"][idx]
        hobs = screen_average_np(h, dk) * sc
        dobs = screen_average_np(dh, dk) * sc
        assert np.max(np.abs(hobs - ref[:, 1]) / np.maximum(np.abs(ref[:, 1]), 1e-300)) < 2e-15, (name, ir)
        assert np.max(np.abs(dobs - ref[:, 2]) / np.maximum(np.abs(ref[:, 2]), 1e-300)) < 2e-15, (name, ir)
