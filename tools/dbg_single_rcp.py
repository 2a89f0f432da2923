#!/usr/bin/env python3
"""Why the single-reciprocal form of the water-table sample, (den - f_z) / (q den), was 2e-6 / 4e-6 off at two far-field fuzz
points while its samples looked as good as any.  For the library named by UCF_LIB_PATH (the product, or
a build with -DUCF_SINGLE_RCP) at the two recorded points (seed 903 set 26, seed 700 set 39):
  1. the stages of the PRODUCTION launch sequence (ucf_debug_stages: level sums, J0-interval areas, totlap) against the
     binary128 oracle: where does the form's noise enter, and how large is it relative to each stage vector;
  2. the same point with tD moved by k x 1e-13 (k = 0..23): every rounding changes, the problem does not -- the spread of the
     result's error against binary128 is the conditioning of the point times the noise of the form, not one realisation."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_hunt
import oracle_lib
from golden_util import load_deck
from unconfined_amd import engine
from unconfined_amd.abi import params_from_deck
O, Q = oracle_lib.Oracle(), oracle_lib.Oracle(quad=True)
c = lambda a: a[..., 0] + 1j * a[..., 1]
vec = lambda got, ref: float((np.abs(got - ref) / np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1e-300)).max())
print("library:", os.environ.get("UCF_LIB_PATH", "product"), "build", engine.build_id())
for seed, nset, tq, rq in ((903, 27, 1.967595600382917, 3.8513397309519557), (700, 40, 7.3769061103986715, 8.669238774305269)):
    for i, bname, ch, tD, rD, zD in fuzz_hunt.sets_of(seed, nset, 320):
        pass
    P = params_from_deck(load_deck(bname)[0].replace(**ch))
    D = O.nondim(P)
    j0z = O.j0_zeros(D.nj0z)
    plan = engine.Plan(P, mode="fast")
    zl = plan.zlay(zD)
    for z0 in range(len(zD)):
        zz, zll = zD[z0:z0 + 1], zl[z0:z0 + 1]
        sv = plan.split_vector(np.array([tq]))
        st = plan.debug_stages(np.array([tq]), sv, np.array([rq]), zz, zll, grid=False)
        hq, dq, sq = Q.point(P, D, j0z, tq, rq, int(sv[0]), zz, zll, stages=True)
        ho, do, so = O.point(P, D, j0z, tq, rq, int(sv[0]), zz, zll, stages=True)
        R = st["R"]
        arg = j0z[sv[0] - 1] / rq
        s = st["state"][0]
        tmp = np.transpose(s[:, :R, :], (1, 2, 0)) * (arg / 2.0); gl = np.transpose(s[:, R + 1:, :], (1, 2, 0))
        print(f"seed {seed} set {nset - 1} depth {z0}: model {P.model} kappa {P.kappa:.3g} fully penetrating {P.d == 0 and P.l == P.b}; layout {st['layout']}")
        print(f"   stage vectors vs binary128, max norm rel. to the vector's largest:  device tmp {vec(tmp, c(sq['tmp'])):.2e} glarea {vec(gl, c(sq['glarea'])):.2e} totlap {vec(st['totlap'][0], c(sq['totlap'])):.2e}"
              f"   | oracle tmp {vec(c(so['tmp']), c(sq['tmp'])):.2e} glarea {vec(c(so['glarea']), c(sq['glarea'])):.2e} totlap {vec(c(so['totlap']), c(sq['totlap'])):.2e}")
        # per interval: the areas alternate in sign and shrink; the error of each relative to the LARGEST area of its Laplace index
        gq, gd, go = c(sq["glarea"])[:, 0, :], gl[:, 0, :], c(so["glarea"])[:, 0, :]
        big = np.abs(gq).max(axis=0)
        print("   area error / largest area of that Laplace index, max over the indices, per J0 interval:  device", np.array2string((np.abs(gd - gq) / big).max(axis=1), precision=1),
              " oracle", np.array2string((np.abs(go - gq) / big).max(axis=1), precision=1))
        print("   |last area| / |first area| (median over the indices):", float(np.median(np.abs(gq[-1]) / np.abs(gq[0]))), " |sum of areas| / |first area|:", float(np.median(np.abs(gq.sum(axis=0)) / np.abs(gq[0]))))
        ks = np.arange(24)
        tds = tq * (1.0 + ks * 1e-13)
        h, _ = plan.drawdown(tds, np.full(len(ks), rq), plan.split_vector(tds), zz, zll)
        ht, _ = Q.batch(P, tds, np.full(len(ks), rq), plan.split_vector(tds), zz, zll, threads=8)
        hr, _ = O.batch(P, tds, np.full(len(ks), rq), plan.split_vector(tds), zz, zll, threads=8)
        e = np.abs(h[:, 0] - ht[:, 0]) / np.abs(ht[:, 0]); er = np.abs(hr[:, 0] - ht[:, 0]) / np.abs(ht[:, 0])
        print(f"   h error vs binary128 over 24 perturbed copies of the point (tD x (1 + k 1e-13)):  device at k=0 {e[0]:.2e}, median {np.median(e):.2e}, max {e.max():.2e}   | reference (oracle) at k=0 {er[0]:.2e}, median {np.median(er):.2e}, max {er.max():.2e}")
