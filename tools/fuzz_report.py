"""summary of a tools/fuzz_hunt.py result: per flagged point the error of fast / faithful / binary64 oracle against the
binary128 evaluation; worst fast-to-oracle error ratios and NaN-pattern mismatches.  usage: fuzz_report.py <json> [n]"""
import json, sys
import numpy as np
d = json.load(open(sys.argv[1]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = []
for r in d["flagged"]:
    for p in r["points"]:
        f, g, o, t = (np.array(p[k]) for k in ("fast", "faithful", "cpu_oracle", "binary128"))
        sc = np.maximum(np.abs(t), 1e-300)
        ef, eg, eo = np.abs(f - t) / sc, np.abs(g - t) / sc, np.abs(o - t) / sc
        ratio = max((a / max(b, 1e-10)) if np.isfinite(a) else 0.0 for a, b in zip(ef, eo))
        rows.append((ratio, r["seed"], r["set"], r["base"], round(r["change"]["kappa"], 4), round(p["rD"], 3), f"{p['tD']:.2e}", r["zLay"],
                     r["nan_pattern_differs_at"], [f"{x:.1e}" for x in ef], [f"{x:.1e}" for x in eg], [f"{x:.1e}" for x in eo]))
rows.sort(key=lambda x: -x[0])
print("sets", d["sets_run"], "flagged", len(d["flagged"]))
print("ratio seed set base kappa rD tD zLay nan_diff | err fast | faithful | oracle")
for r in rows[:top]:
    print(round(r[0], 1), *r[1:])
print("NaN-pattern mismatches:", [r[1:9] for r in rows if r[8] > 0])
