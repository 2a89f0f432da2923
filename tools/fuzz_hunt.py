#!/usr/bin/env python3
"""Hunt for the parameter sets on which the two flavours part ways (NaN patterns, or values by more than the
reference's own distance from exact arithmetic), keep them with everything needed to replay them, and say for each
which evaluator hand-over is involved.  usage: tools/fuzz_hunt.py [nsets per seed] [seed0] [nseeds] [out.json]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_deck
from unconfined_amd.abi import params_from_deck

BASES = {1: "hantush_lay1", 3: "c3_moench", 4: "malama_fullpen", 5: "neuman74_partpen", 6: "mishra_fd30"}


def sets_of(seed, nsets, npts=320):
    """the deterministic stream of (index, base deck name, changed deck fields, tD, rD, zD) of one seed"""
    rng = np.random.default_rng(seed)
    base = {m: load_deck(n)[0] for m, n in BASES.items()}
    for i in range(nsets):
        model = int(rng.choice(list(base)))
        dk = base[model]
        b = dk.b
        full = rng.random() < 0.3
        d = 0.0 if full else b * rng.uniform(0.0, 0.4)
        l = b if full else min(b, d + b * rng.uniform(0.1, 0.6))
        ch = dict(Kr=dk.Kr * 10 ** rng.uniform(-1, 1), kappa=10 ** rng.uniform(-1.5, 0.3), Ss=dk.Ss * 10 ** rng.uniform(-1, 1),
                  Sy=min(0.45, dk.Sy * 10 ** rng.uniform(-0.7, 0.3)), l=l, d=d, beta=(0.0 if rng.random() < 0.5 else 10 ** rng.uniform(-2, 1)))
        tD = 10.0 ** rng.uniform(-2, 5, npts); rD = 10.0 ** rng.uniform(-1, 1, npts)
        zD = np.sort(rng.uniform(0.02, 0.98, 2))
        yield i, BASES[model], ch, tD, rD, zD


def main():
    from unconfined_amd import engine
    import oracle_lib
    nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    nseeds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "fuzz_flagged.json")
    npts = int(os.environ.get("UCF_FUZZ_NPTS", "320"))
    oracle, oracle_q = oracle_lib.Oracle(), oracle_lib.Oracle(quad=True)
    flagged, total = [], 0
    for seed in range(seed0, seed0 + nseeds):
        for i, bname, ch, tD, rD, zD in sets_of(seed, nsets, npts):
            dk2 = load_deck(bname)[0].replace(**ch)
            P = params_from_deck(dk2)
            try:
                pf, pg = engine.Plan(P, mode="fast"), engine.Plan(P, mode="faithful")
            except Exception:
                continue
            zl = pf.zlay(zD)
            sv = pf.split_vector(tD)
            hf, dhf = pf.drawdown(tD, rD, sv, zD, zl)
            hg, dhg = pg.drawdown(tD, rD, sv, zD, zl)
            total += 1
            sc = max(np.nanmax(np.abs(hg)), 1e-300)
            e = np.abs(hf - hg) / np.maximum(np.abs(hg), 1e-4 * sc)
            nan_diff = np.argwhere(np.isnan(hf) != np.isnan(hg))
            k = np.unravel_index(np.nanargmax(e), e.shape)
            if len(nan_diff) == 0 and not float(e[k]) > 1e-6:
                continue
            pts = sorted(set([int(k[0])] + [int(q[0]) for q in nan_diff[:4]]))
            ho, _ = oracle.batch(P, tD[pts], rD[pts], sv[pts], zD, zl)
            ht, _ = oracle_q.batch(P, tD[pts], rD[pts], sv[pts], zD, zl, threads=16)
            rec = {"seed": seed, "set": i, "base": bname, "change": {a: float(v) for a, v in ch.items()}, "zD": zD.tolist(), "zLay": zl.tolist(),
                   "nan_pattern_differs_at": len(nan_diff), "worst_rel_diff": float(e[k]), "points": []}
            for j, q in enumerate(pts):
                rec["points"].append({"index": int(q), "tD": float(tD[q]), "rD": float(rD[q]), "sv": int(sv[q]), "fast": hf[q].tolist(), "faithful": hg[q].tolist(),
                                      "cpu_oracle": ho[j].tolist(), "binary128": ht[j].tolist()})
            flagged.append(rec)
            print("flagged", seed, i, bname, "kappa %.3g" % ch["kappa"], "nan-diff", len(nan_diff), "rel", "%.2e" % float(e[k]), flush=True)
    json.dump({"sets_run": total, "npts": npts, "flagged": flagged}, open(out, "w"), indent=1)
    print("sets", total, "flagged", len(flagged))


if __name__ == "__main__":
    main()
