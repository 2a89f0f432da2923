"""The kernel instantiations that actually run, checked between "sample" and "h, dh" (SURVEY.md section 8c fixture 2).

ucf_debug_stages runs the PRODUCTION launch sequence for the sizes it is given -- the launcher's own choice of lane layout,
FOLD / LAY3 / register budget / constants-in-VGPRs instantiation -- and reads back what those kernels left in the
workspace: per Laplace sample the level sums of the tanh-sinh part, the Gauss-Lobatto areas between J0 zeros, and the
accelerated transform totlap.  They are compared with tests/golden/midstages.npz (oracle/gen_midstages.py: the oracle's
loop body, driver.f90:129-216) for C2 (fully penetrating: FOLD), C2pp, C3 (two depths, 3 waves/SIMD), C4 and C5, in the
three lane layouts: 0 lane = Laplace sample (a short list), 1 lane = time (the grid: what bench.py runs), 3 lane = point
(a long list).  ucf_debug_wynn / ucf_debug_dehoog_tiles run the register-resident epsilon table (wynn_regs<12>) and the
tiled de Hoog kernel that finish those launch sequences, in both flavours.

Bars (written where they are used): a stage vector over the 2M+1 Laplace samples is compared in the max norm relative to
the vector's own max modulus; the faithful flavour differs from the oracle by the device libm only, the fast flavour by
its table-driven primitives and FMA contraction as well; both must be as close to the binary128 evaluation as the
binary64 oracle is, within a small factor."""
import os

import numpy as np
import pytest

from golden_util import GOLD, bits_equal, load_deck

pytestmark = pytest.mark.gpu

DECKS = ["c2_neuman74_fullpen", "neuman74_partpen", "c3_moench", "c4_malama_partpen", "c5_mishra_fd64"]
MODES = ["faithful", "fast"]
REPORT = {}


@pytest.fixture(scope="module")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unconfined_amd import engine as e
    return e


@pytest.fixture(scope="module")
def mid():
    return np.load(os.path.join(GOLD, "midstages.npz"))


def _cplx(a):
    return a[..., 0] + 1j * a[..., 1]


def _vec_err(got, ref):
    """max over the Laplace samples (last axis) of |got - ref|, relative to the largest modulus of ref along it"""
    scale = np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1e-300)
    return float((np.abs(got - ref) / scale).max())


def _grid(oracle, mid, name):
    dk, ts, P = load_deck(name)
    D = oracle.nondim(P)
    nt = int(mid["nt"][0])
    tD = 10.0 ** oracle.linspace(-2.0, 4.0, nt)
    sv = oracle.split_vector(list(dk.j0s), tD)
    zz = oracle.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd)
    zD = zz / D.Lc
    return dk, P, D, tD, sv, np.array(mid["radii"]), zD, oracle.zlay(D, zD)


def _check_point(name, mode, layout, k, st, q, arg, mid, bars):
    """device stages of point q against fixture point k"""
    R, nacc = st["R"], st["nacc"]
    ref_tmp, ref_gl, ref_tl = _cplx(mid[f"{name}_{k}_tmp"]), _cplx(mid[f"{name}_{k}_glarea"]), _cplx(mid[f"{name}_{k}_totlap"])
    q_tmp, q_gl, q_tl = _cplx(mid[f"{name}_{k}_tmp_q"]), _cplx(mid[f"{name}_{k}_glarea_q"]), _cplx(mid[f"{name}_{k}_totlap_q"])
    out = {}
    if st["has_state"]:
        assert (st["ndone"][q] == st["ndone"][q][0]).all()
        s = st["state"][q]                                   # [np, R+1+nacc, nz]
        tmp = np.transpose(s[:, :R, :], (1, 2, 0)) * (arg / 2.0)          # [R, nz, np]   (driver.f90:135,154)
        gl = np.transpose(s[:, R + 1:, :], (1, 2, 0))                     # [nacc, nz, np]
        out["tmp"] = (_vec_err(tmp, ref_tmp), _vec_err(tmp, q_tmp), _vec_err(ref_tmp, q_tmp))
        out["glarea"] = (_vec_err(gl, ref_gl), _vec_err(gl, q_gl), _vec_err(ref_gl, q_gl))
    tl = st["totlap"][q]                                     # [nz, np]
    out["totlap"] = (_vec_err(tl, ref_tl), _vec_err(tl, q_tl), _vec_err(ref_tl, q_tl))
    for key, (e_ref, e_q, ref_q) in out.items():
        ent = REPORT.setdefault(key, {}).setdefault(f"{name}/{mode}/L{layout}", [0.0, 0.0, 0.0])
        ent[0], ent[1], ent[2] = max(ent[0], e_ref), max(ent[1], e_q), max(ent[2], ref_q)
        vs_oracle, vs_truth = bars[key]
        # against the binary64 oracle, OR (where the oracle itself is that far from exact arithmetic: cancellation in the
        # series) as close to the binary128 evaluation as the oracle is, times a small factor
        assert e_ref <= vs_oracle or e_q <= vs_truth * max(ref_q, 1e-15), (name, mode, layout, k, key, e_ref, e_q, ref_q)
    return out


# stage vectors relative to their max modulus over the Laplace samples: (bar against the oracle, factor on the oracle's own
# distance from binary128).  Level sums and areas are sums of <= 63 / 48 samples that are each good to a few ulp; totlap has been
# through Neville (h -> 0) and Wynn-epsilon, which amplify
BARS = {"faithful": {"tmp": (2e-13, 8.0), "glarea": (2e-13, 8.0), "totlap": (1e-10, 16.0)},
        "fast": {"tmp": (5e-13, 16.0), "glarea": (5e-13, 16.0), "totlap": (1e-10, 32.0)}}


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", DECKS)
def test_stages_lane_time_layout(engine, oracle, mid, name, mode):
    """LAYOUT 1 (the grid entry: integrate_kernel<F,1,W,...> / integrate_generic_kernel<F,1> -> finish_kernel<1,...>)"""
    dk, P, D, tD, sv, rD, zD, zl = _grid(oracle, mid, name)
    plan = engine.Plan(P, mode=mode)
    st = plan.debug_stages(tD, sv, rD, zD, zl, grid=True)
    assert st["layout"] == 1
    j0z = oracle.j0_zeros(D.nj0z)
    # the grid entry itself gives these h, dh
    hg, dg = plan.drawdown_grid(tD, sv, rD, zD, zl)
    assert np.array_equal(st["h"].reshape(hg.shape), hg, equal_nan=True) and np.array_equal(st["dh"].reshape(dg.shape), dg, equal_nan=True)
    for k, (it, ir) in enumerate(mid["picks"]):
        _check_point(name, mode, 1, k, st, it * len(rD) + ir, j0z[sv[it] - 1] / rD[ir], mid, BARS[mode])


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", DECKS)
def test_stages_lane_point_and_lane_sample_layouts(engine, oracle, mid, name, mode):
    """LAYOUT 3 (a list of 256 points, ordered by radius) and LAYOUT 0 (the five fixture points on their own)"""
    dk, P, D, tD, sv, rD, zD, zl = _grid(oracle, mid, name)
    plan = engine.Plan(P, mode=mode)
    j0z = oracle.j0_zeros(D.nj0z)
    nt = len(tD)
    tl = np.tile(tD, len(rD)); rl = np.repeat(rD, nt); sl = np.tile(sv, len(rD))          # point = ir * nt + it
    st = plan.debug_stages(tl, sl, rl, zD, zl, grid=False)
    assert st["layout"] == (3 if st["has_state"] else 0)
    for k, (it, ir) in enumerate(mid["picks"]):
        _check_point(name, mode, st["layout"], k, st, ir * nt + it, j0z[sv[it] - 1] / rD[ir], mid, BARS[mode])
    order = np.argsort([rD[ir] for it, ir in mid["picks"]], kind="stable")
    pk = [tuple(mid["picks"][i]) for i in order]
    t5 = np.array([tD[it] for it, ir in pk]); r5 = np.array([rD[ir] for it, ir in pk]); s5 = np.array([sv[it] for it, ir in pk], np.int32)
    st0 = plan.debug_stages(t5, s5, r5, zD, zl, grid=False)
    assert st0["layout"] == 0
    for q, i in enumerate(order):
        it, ir = mid["picks"][i]
        _check_point(name, mode, 0, int(i), st0, q, j0z[sv[it] - 1] / rD[ir], mid, BARS[mode])


def test_wynn_epsilon_in_registers(engine):
    """wynn_regs<12> (what finish_kernel runs) on the reference's fixtures: the faithful flavour bit for bit incl. truncation
    at the first non-finite term, the -999999.9 sentinel and the absolute-epsilon early exit; the fast flavour (squared
    modulus test, reciprocal by Newton) takes the same exits and agrees to a few ulp of the result's modulus"""
    z = np.load(os.path.join(GOLD, "stages_generic.npz"))
    nw = int(z["counts"][0])
    by_n = {}
    for i in range(nw):
        n = len(z[f"wynn_in_{i}"])
        if n <= 12:
            by_n.setdefault(n, []).append(i)
    assert sum(len(v) for v in by_n.values()) >= 4
    seen = set()
    for n, idx in by_n.items():
        ser = np.stack([z[f"wynn_in_{i}"] for i in idx])
        ref = np.stack([z[f"wynn_out_{i}"] for i in idx])
        acc, st = engine.debug_wynn(ser, "faithful")
        acc_l, st_l = engine.wynn_epsilon(ser)
        seen |= set(int(s) for s in st)
        assert np.array_equal(st, st_l)
        for k in range(len(idx)):
            assert bits_equal(acc[k], ref[k]), ("wynn_regs faithful", idx[k])
        accf, stf = engine.debug_wynn(ser, "fast")
        assert np.array_equal(stf, st), (stf, st)
        fin = np.isfinite(ref).all(axis=1)
        err = np.abs(_cplx(accf) - _cplx(ref))[fin] / np.maximum(np.abs(_cplx(ref))[fin], 1e-300)
        assert err.max(initial=0.0) <= 1e-13, float(err.max())
    assert {0, 1, 2, 3} <= seen | {0, 1, 2, 3} and len(seen) >= 3
    # random convergent / oscillating series of every length the plans use (nacc = 4 .. 12)
    rng = np.random.default_rng(12)
    for n in (4, 7, 10, 12):
        k = np.arange(n)
        ser = np.stack([np.stack([(-0.7) ** k * rng.uniform(0.5, 1.5) / (1 + k) ** rng.uniform(0.5, 2), 0.3 * (-0.6) ** k * rng.standard_normal() / (1 + k)], axis=1)
                        for _ in range(200)])
        a0, s0 = engine.wynn_epsilon(ser)
        a1, s1 = engine.debug_wynn(ser, "faithful")
        a2, s2 = engine.debug_wynn(ser, "fast")
        assert np.array_equal(a0, a1) and np.array_equal(s0, s1)
        assert np.array_equal(s0, s2)
        assert (np.abs(_cplx(a2) - _cplx(a0)) / np.abs(_cplx(a0))).max() <= 1e-12


def test_dehoog_tiled_kernel(engine, oracle, oracle_quad):
    """dehoog_tiles_kernel (cooperative quotient-difference rhombus + one continued fraction per lane) against the
    reference's fixtures and the oracle: value within 1e-13 (faithful; same bits as the wave-cooperative hook ucf_dehoog,
    whose arithmetic it repeats) / 5e-12 (fast: unscaled quotient) where the inversion is well conditioned, the derivative
    channel p F(p) likewise"""
    z = np.load(os.path.join(GOLD, "stages_generic.npz"))
    nd = int(z["counts"][2])
    done = 0
    for i in range(nd):
        M, alpha, tol, t, tee = z[f"dehoog_par_{i}"]
        if abs(tee - 2.0 * t) > 1e-15 * tee or 2 * int(M) + 1 > 64:
            continue                                     # the kernel runs the driver's T = 2 t (driver.f90:106), 2M+1 <= 64
        fp = z[f"dehoog_fp_{i}"]
        ref = float(z[f"dehoog_out_{i}"][0])
        # (fast: q e / e' by an unscaled Newton reciprocal instead of the scaled division: the worst fixture vector -- a
        #  table with e' ~ 1e-9 -- moves by 1.2e-12)
        for mode, bar in (("faithful", 1e-13), ("fast", 5e-12)):
            h, dh = engine.debug_dehoog_tiles(int(M), alpha, tol, [t], fp[None], mode)
            if np.isnan(ref):
                assert np.isnan(h[0])
            elif ref == 0.0:
                assert h[0] == 0.0
            else:
                assert abs(h[0] - ref) <= bar * abs(ref), (i, mode, h[0], ref)
        done += 1
    assert done >= 3
    # many vectors at once (tiles of 4, ragged tail), smooth transforms F(p) = 1/(p+1)^2 [f = t e^-t] and 1/sqrt(p) e^{-1/p}-like decay
    M, alpha, tol = 26, 1e-8, 1e-9
    t = 10.0 ** np.linspace(-1, 1.3, 23)
    fps = []
    for tt in t:
        p = oracle.pvalues(2 * tt, M, alpha, tol)
        pc = p[:, 0] + 1j * p[:, 1]
        F = 1.0 / (pc + 1.0) ** 2 + 0.3 / (pc + 0.2)
        fps.append(np.stack([F.real, F.imag], axis=1))
    fps = np.array(fps)
    want = t * np.exp(-t) + 0.3 * np.exp(-0.2 * t)
    for mode, bar in (("faithful", 2e-13), ("fast", 5e-12)):
        h, dh = engine.debug_dehoog_tiles(M, alpha, tol, t, fps, mode)
        ho = np.array([oracle.dehoog(M, alpha, tol, tt, 2 * tt, fps[i]) for i, tt in enumerate(t)])
        assert (np.abs(h - ho) / np.abs(ho)).max() <= bar, (mode, float((np.abs(h - ho) / np.abs(ho)).max()))
        assert (np.abs(h - want) / want).max() < 1e-7
        pcs = [oracle.pvalues(2 * tt, M, alpha, tol) for tt in t]
        do = np.array([oracle.dehoog(M, alpha, tol, tt, 2 * tt, np.stack([(_cplx(fps[i]) * _cplx(pcs[i])).real, (_cplx(fps[i]) * _cplx(pcs[i])).imag], axis=1)) * tt
                       for i, tt in enumerate(t)])
        assert (np.abs(dh - do) / np.maximum(np.abs(do), 1e-3)).max() <= 50 * bar, (mode, float((np.abs(dh - do) / np.maximum(np.abs(do), 1e-3)).max()))
    hf, _ = engine.debug_dehoog_tiles(M, alpha, tol, t, fps, "faithful")
    hw = np.array([engine.dehoog(M, alpha, tol, tt, 2 * tt, fps[i][None])[0] for i, tt in enumerate(t)])
    assert np.array_equal(hf, hw)              # the tiled kernel repeats dehoog_wave's arithmetic: same bits (faithful)


def test_zz_stage_report():
    """(last in this file) keep the worst stage errors per deck x flavour x layout next to the parity report"""
    import json
    if not REPORT:
        pytest.skip("no stage test ran")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "stages_r03.json"), "w") as f:
        json.dump({"what": "worst max-norm error of a stage vector over its Laplace samples, relative to the vector's max modulus: "
                           "[device vs oracle, device vs binary128, oracle vs binary128]", "stages": REPORT}, f, indent=1, sort_keys=True)
