"""Pumping schedules beyond the two-parameter types (reference time.f90:81-122): the piecewise-linear rate in its
in-bounds reading (include/ucf.h) and the 100-step piecewise-constant rate, which the reference
cannot even read (it sizes the record with mod(type,100), driver_io.f90:124).

CPU part: the oracle's Laplace-domain multiplier equals the transform of the rate function the header describes,
integrated segment by segment in closed form (an independent derivation), and tends to the step response as a ramp
gets steep.  GPU part: the device against the oracle and the binary128 evaluation."""
import numpy as np
import pytest

from golden_util import load_deck, rel_err
from unconfined_amd.abi import params_from_deck


def _rate_transform(knots, tf, rates, p):
    """L{Q}(p) for Q = 0 up to knots[0], linear through (knots[k], y_k) with y_0 = 0, y_k = rates[k-1] at knots[k]
    (k >= 1) and rates[-1] at tf, constant after tf -- each linear piece integrated exactly"""
    t = list(knots) + [tf]
    y = [0.0] + list(rates)
    out = np.zeros_like(p)
    for k in range(len(t) - 1):
        a, b = t[k], t[k + 1]
        w = (y[k + 1] - y[k]) / (b - a)
        # int_a^b (y_k + w (s - a)) e^{-p s} ds
        ea, eb = np.exp(-p * a), np.exp(-p * b)
        out = out + y[k] * (ea - eb) / p + w * ((ea - eb) / p ** 2 - (b - a) * eb / p)
    return out + y[-1] * np.exp(-p * tf) / p


def _laptime(oracle, dk, p):
    """the oracle's time multiplier: ratio of the Theis sample with this schedule to the one with a unit step at t = 0"""
    P = params_from_deck(dk)
    D = oracle.nondim(P)
    pa = np.stack([p.real, p.imag], axis=1)
    zD = np.array([0.5]); zl = oracle.zlay(D, zD)
    f = oracle.soln(P, D, 0.7, 0.4, pa, zD, zl)[0]
    P1 = params_from_deck(dk.replace(timeType=1, timePar=[0.0, 1.0]))
    g = oracle.soln(P1, D, 0.7, 0.4, pa, zD, zl)[0]
    return (f[:, 0] + 1j * f[:, 1]) / (g[:, 0] + 1j * g[:, 1]) / p          # step multiplier is 1/p


@pytest.mark.parametrize("quad", [False, True])
def test_piecewise_linear_multiplier_is_the_transform_of_the_documented_rate(oracle, oracle_quad, quad):
    O = oracle_quad if quad else oracle
    dk, ts, P = load_deck("c1_theis")
    p = np.array([0.3 + 0.0j, 1.0 + 2.0j, 0.05 + 7.0j, 4.0 - 1.0j])
    for knots, tf, rates in (([0.5], 2.0, [3.0]), ([0.0, 1.0, 2.5], 4.0, [1.0, 0.2, 0.7]), ([1.0, 1.5, 3.0, 3.5, 6.0], 9.0, [2.0, 2.0, 0.0, 1.0, 0.5])):
        n = len(knots)
        d = dk.replace(timeType=-(100 + n), timePar=list(knots) + [tf] + list(rates))
        got = _laptime(O, d, p)
        want = _rate_transform(knots, tf, rates, p)
        assert np.abs(got - want).max() <= 2e-13 * np.abs(want).max(), (knots, np.abs(got - want).max())
    # a steep ramp is a step: 0 -> y within 1e-7 time units at t1
    d = dk.replace(timeType=-101, timePar=[0.5, 0.5 + 1e-7, 2.5])
    got = _laptime(O, d, p)
    step = 2.5 * np.exp(-0.5 * p) / p
    assert np.abs(got - step).max() <= 1e-5 * np.abs(step).max()


def test_hundred_step_schedule_equals_its_short_form(oracle):
    """timeType = -100: 100 steps of which only 3 change the rate == the 3-step schedule (same increments)"""
    dk, ts, P = load_deck("c1_theis")
    ti = list(np.linspace(0.0, 9.9, 100)); q = [1.0] * 40 + [2.5] * 30 + [0.5] * 30
    long = dk.replace(timeType=-100, timePar=ti + [12.0] + q)
    short = dk.replace(timeType=-3, timePar=[ti[0], ti[40], ti[70], 12.0, 1.0, 2.5, 0.5])
    p = np.array([0.3 + 0.0j, 1.0 + 2.0j, 0.05 + 7.0j])
    a, b = _laptime(oracle, long, p), _laptime(oracle, short, p)
    assert np.abs(a - b).max() <= 1e-13 * np.abs(b).max()
    # the deck writer / parser round-trips 201 parameters
    from unconfined_amd.deck import Deck
    import os, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "long.in")
        long.write(path)
        back = Deck.read(path)
    assert back.timeType == -100 and len(back.timePar) == 201 and back.timePar[150] == long.timePar[150]


def test_coincident_knots_are_refused_like_the_reference():
    """time.f90:110-112 stops on a vertical segment; the ABI returns an error instead (no GPU needed to say so)"""
    from unconfined_amd import engine
    from unconfined_amd.lib import UcfError
    dk, ts, P = load_deck("c1_theis")
    bad = params_from_deck(dk.replace(timeType=-102, timePar=[1.0, 1.0, 3.0, 1.0, 2.0]))
    with pytest.raises(UcfError) as e:
        engine.Plan(bad)
    assert "vertical" in str(e.value)
    with pytest.raises(UcfError):
        engine.Plan(params_from_deck(dk.replace(timeType=-201, timePar=[0.0] * 3)))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["faithful", "fast"])
@pytest.mark.parametrize("base,sched", [("c1_theis", "linear3"), ("neuman74_partpen", "linear5"), ("c1_theis", "const100")])
def test_schedules_device_vs_oracle(oracle, oracle_quad, base, sched, mode):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unconfined_amd import engine
    dk, ts, P0 = load_deck(base)
    Tc = oracle.nondim(P0).Tc
    if sched == "linear3":
        tp = [0.0, 20.0, 50.0, 80.0, 1.0, 0.2, 0.7]; tt = -103
    elif sched == "linear5":
        tp = [0.0, 5.0, 30.0, 31.0, 60.0, 90.0, 2.0, 2.0, 0.0, 1.0, 0.5]; tt = -105
    else:
        ti = list(np.linspace(0.0, 99.0, 100)); tp = ti + [150.0] + [1.0 + 0.5 * np.sin(0.3 * k) for k in range(100)]; tt = -100
    # (timePar is used with the dimensionless p without scaling, SURVEY.md quirk Q3: times here are dimensionless)
    tp = [v / Tc if i <= (len(tp) - 1) // 2 else v for i, v in enumerate(tp)]
    d = dk.replace(timeType=tt, timePar=tp)
    P = params_from_deck(d)
    plan = engine.Plan(P, mode=mode)
    D = plan.derived
    t = 10.0 ** np.linspace(-1, 3.5, 24)
    tD = t / D.Tc; rD = np.full(len(t), 85.1 / D.Lc if base != "c1_theis" else 0.5); sv = plan.split_vector(tD)
    zD = np.array([0.6]); zl = plan.zlay(zD)
    h, dh = plan.drawdown(tD, rD, sv, zD, zl)
    ho, dho = oracle.batch(P, tD, rD, sv, zD, zl)
    ht, dht = oracle_quad.batch(P, tD, rD, sv, zD, zl, threads=8)
    floor = 1e-3 / (1.0 if d.dimless else D.Hc)
    assert np.isfinite(h).all() and np.abs(ho).max() > 10 * floor
    for got, ref, truth, label in ((h, ho, ht, "h"), (dh, dho, dht, "dh")):
        eg, er = rel_err(got, truth, floor), rel_err(ref, truth, floor)
        assert eg.max() <= max(1e-10, 20.0 * er.max()), (label, float(eg.max()), float(er.max()))
