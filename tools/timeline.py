#!/usr/bin/env python3
"""Diagnostic (GPU box, library built with -DUCF_TIMELINE: tools/ubench/build_variant.sh tl -DUCF_TIMELINE): when and where every
work item of a lane = time launch ran.  usage: UCF_LIB_PATH=tools/ubench/libucf_tl.so python tools/timeline.py [nt=128] [nr=256]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from unconfined_amd import engine
from unconfined_amd.abi import params_from_deck
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 256
wl = os.environ.get("UCF_TL_WORKLOAD", "c2")
dk = bench.workload_deck(wl)[0]
plan = engine.Plan(params_from_deck(dk), mode="fast")
D = plan.derived
tD = engine.logspace(-1, 8, nt) / D.Tc
rD = 10.0 ** engine.linspace(-1.0, 1.0, nr)
zD = engine.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd) / D.Lc; zl = plan.zlay(zD)
sv = plan.split_vector(tD)
st = plan.debug_stages(tD, sv, rD, zD, zl, grid=True)      # (UCF_DEBUG_REPS=8: the eighth of eight launches back to back)
R = st["R"]
slot = st["state"][:, :, R, 0]                       # [npts, np] complex: lane 0 of each tile carries (t0, t1), lane 1 (hwid, xcc)
pts = np.arange(nt * nr).reshape(nt, nr)
l0 = pts[0::64].ravel(); l1 = pts[1::64].ravel()
t0 = slot[l0].real.copy().view(np.int64); t1 = slot[l0].imag.copy().view(np.int64)
hw = slot[l1].real.astype(np.int64); xcc = slot[l1].imag.astype(np.int64) & 0xf
t0 = t0.ravel().astype(float); t1 = t1.ravel().astype(float); hw = hw.ravel(); xcc = xcc.ravel()
np.savez("gpurun_out/timeline.npz", t0=t0, t1=t1, hw=hw, xcc=xcc)
# s_memtime is not synchronised across the chip (CU pairs share a counter): one CU at a time
se = (hw >> 13) & 7; cu = (hw >> 8) & 0xf
key = xcc * 10000 + se * 100 + cu
keys = np.unique(key)
spans, nitems = [], []
for k in keys:
    m = key == k
    spans.append(t1[m].max() - t0[m].min()); nitems.append(int(m.sum()))
spans = np.array(spans)
dur = t1 - t0
print(f"{len(keys)} CUs; items per CU {min(nitems)}..{max(nitems)}; busy span per CU kcycles: min {spans.min() / 1e3:.0f} median {np.median(spans) / 1e3:.0f} max {spans.max() / 1e3:.0f}; "
      f"item duration median {np.median(dur) / 1e3:.0f} p10 {np.percentile(dur, 10) / 1e3:.0f} p90 {np.percentile(dur, 90) / 1e3:.0f}")
for k in keys[[0, len(keys) // 2, -1]]:
    m = key == k
    a, b = t0[m] - t0[m].min(), t1[m] - t0[m].min()
    step = 250e3
    edges = np.arange(0, b.max() + step, step)
    infl = [int(((a < e2) & (b > e1)).sum()) for e1, e2 in zip(edges[:-1], edges[1:])]
    o = np.argsort(a)
    print(f"CU {k}: {int(m.sum())} items; in flight per 250-kcycle bin {infl}")
    print(f"   first starts (kcycles) {np.round(a[o][:30] / 1e3).astype(int).tolist()}")
    print(f"   last ends   (kcycles) {np.round(np.sort(b)[-30:] / 1e3).astype(int).tolist()}")
