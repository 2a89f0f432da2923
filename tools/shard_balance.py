#!/usr/bin/env python3
"""How evenly does a contiguous block partition of a sweep spread the work?  Times the grid call of each of G
blocks of time rows and of each of G blocks of radii (one GPU, one block after the other).
usage: tools/shard_balance.py [workload] [G]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import workload_deck
from unconfined_amd import engine
from unconfined_amd.abi import params_from_deck

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dk, nt, nr, name = workload_deck(wl)
if wl in ("c3", "c4"):
    nt, nr = nt // 4, nr // 4
plan = engine.Plan(params_from_deck(dk), mode="fast")
D = plan.derived
tD = engine.logspace(-1, 8, nt) / D.Tc
sv = plan.split_vector(tD)
rD = 10.0 ** engine.linspace(-1.0, 1.0, nr)
zD = engine.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd) / D.Lc
zl = plan.zlay(zD)
dev = torch.device("cuda:0")
d_t = torch.tensor(tD, device=dev); d_r = torch.tensor(rD, device=dev); d_s = torch.tensor(sv.astype(np.int32), device=dev)
out = torch.zeros(2, nt * nr * len(zD), dtype=torch.float64, device=dev)
s = torch.cuda.current_stream()

def run(t0, t1, r0, r1, reps=3):
    best = 1e9
    for _ in range(reps + 1):
        torch.cuda.synchronize(); a = time.perf_counter()
        plan.drawdown_grid_device(t1 - t0, d_t[t0:].data_ptr(), d_s[t0:].data_ptr(), r1 - r0, d_r[r0:].data_ptr(), zD, zl,
                                  out[0].data_ptr(), out[1].data_ptr(), stream=s.cuda_stream)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - a)
    return best * 1e3

full = run(0, nt, 0, nr)
rows = [run(g * nt // G, (g + 1) * nt // G, 0, nr) for g in range(G)]
cols = [run(0, nt, g * nr // G, (g + 1) * nr // G) for g in range(G)]
# cyclic tiles of 64 times (what a balanced partition would see): block g takes tiles g, g+G, ...
print(json.dumps({"workload": name, "nt": nt, "nr": nr, "G": G, "full_ms": full,
                  "time_blocks_ms": rows, "radius_blocks_ms": cols,
                  "time_blocks_max_over_mean": max(rows) / (sum(rows) / G), "radius_blocks_max_over_mean": max(cols) / (sum(cols) / G),
                  "strong_speedup_time_blocks": full / max(rows), "strong_speedup_radius_blocks": full / max(cols)}))
