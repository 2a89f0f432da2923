#!/usr/bin/env python3
"""Copy the evidence of tools/gpu_final.sh from gpurun_out/ (scratch) into profiles/ (tracked):
  profiles/<tag>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  profiles/<tag>_bench_<...>.json              bench lines
  profiles/pmc_r03.json                        per workload and kernel: counters per launch, executed fp64 flop, VALU-busy,
                                               HBM bytes -- each entry stamped with the build id of the library that ran
                                               (bench.py only uses entries whose build id is its own library's)
usage: tools/collect_profiles.py [tag]"""
import collections, csv, glob, json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
WORKLOADS = ("c2", "c2pp", "c3", "c4", "c5", "c1", "hstorage", "mnm")


def norm(name):
    """rocprofv3 kernel name -> the name the library's timers use: no 'void ', no argument list"""
    name = name.strip().strip('"')
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):           # cut at the '(' that opens the argument list (template args may hold parentheses)
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def bench_line(log):
    try:
        for ln in reversed(open(log).read().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
    except Exception:
        pass
    return None


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    files = sorted(glob.glob(os.path.join(G, d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:                       # newest run only (gpurun merges, it does not clean)
        for r in csv.DictReader(open(f)):
            out[norm(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()}


prof_path = os.path.join(P, "pmc_r03.json")
try:
    prof = json.load(open(prof_path))
except Exception:
    prof = {}
prof.setdefault("workloads", {})
prof["what"] = ("rocprofv3 --pmc passes of tools/gpu_final.sh over `python3 bench.py --steps 1 --warmup 1 --no-cpu --workload W` (fast flavour, one "
                "MI355X, full-size sweeps); averages per dispatch.  fp64_flop_per_launch = 64 lanes x (2 FMA + ADD + MUL + TRANS) wave instructions; "
                "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB "
                "(MI355X_MICROARCH.md: FETCH_SIZE reports half of wide coalesced reads on gfx950; separate passes, the TCC counters do not fit one); "
                "*_per_wave_abscissa = instructions / work items of the launch (64 times x radius x Laplace sample) / abscissae of the workload (integrate kernels only; "
                "a work item is one wave except in the last round of a launch, whose items run in 8 parts)")
for W in WORKLOADS:
    ks = sorted(glob.glob(os.path.join(G, f"fin_{W}_trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], os.path.join(P, f"{tag}_{W}_kernel_stats.csv"))
    line = bench_line(os.path.join(G, f"fin_{W}_sq.log")) or bench_line(os.path.join(G, f"fin_{W}_trace.log"))
    sq, f64 = counters(f"fin_{W}_sq"), counters(f"fin_{W}_f64")
    fe, wr = counters(f"fin_{W}_FETCH_SIZE"), counters(f"fin_{W}_WRITE_SIZE")
    if not line or not sq:
        continue
    nabs = int(re.search(r"\((\d+) abscissae", line["config"]["workload"]).group(1))
    # work items of one launch of an integrate kernel (lane = time layout of the bench sweeps: 64 times x one radius x one
    # Laplace sample, all depths).  NOT SQ_WAVES: the last round of a launch runs its items in several parts, one wave each
    M_ = int(re.search(r"M=(\d+)", line["config"]["workload"]).group(1))
    items = line["roofline"]["points_per_kernel_launch"] / 64.0 * (2 * M_ + 1)
    kern = {}
    for k in sorted(set(sq) | set(f64) | set(fe) | set(wr)):
        if "ucf_" not in k:
            continue
        c = dict(sq.get(k, {})); c.update(f64.get(k, {}))
        e = {"counters_per_launch": c}
        if "SQ_INSTS_VALU_FMA_F64" in c:
            arith = c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]
            e["fp64_flop_per_launch"] = 64.0 * (arith + c["SQ_INSTS_VALU_FMA_F64"])
            if "integrate" in k and c.get("SQ_WAVES"):
                e["fp64_arith_per_wave_abscissa"] = arith / items / nabs
        # (only for the long kernels: GRBM_GUI_ACTIVE of a 2 ms kernel is dominated by ramp-up and drain and the ratio overshoots 1)
        if c.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in c and ("integrate" in k or "point_kernel" in k):
            e["valu_busy"] = min(1.0, c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024))
        if "integrate" in k and c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c:
            e["valu_per_wave_abscissa"] = c["SQ_INSTS_VALU"] / items / nabs
            e["salu_per_wave_abscissa"] = c["SQ_INSTS_SALU"] / items / nabs
            e["work_items_per_launch"], e["waves_per_launch"] = items, c["SQ_WAVES"]
        if k in fe or k in wr:
            f_, w_ = fe.get(k, {}).get("FETCH_SIZE", 0.0), wr.get(k, {}).get("WRITE_SIZE", 0.0)
            e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"] = f_, w_
            e["hbm_bytes_per_launch"] = (2 * f_ + w_) * 1024
        kern[k] = e
    mode = line["config"]["mode"]
    prof["workloads"].setdefault(W, {})[mode] = {
        "build_id": line["config"]["build_id"], "points_per_launch": line["roofline"]["points_per_kernel_launch"],
        "launches_per_step": line["roofline"].get("kernel_launches_per_step"), "workload": line["config"]["workload"], "kernels": kern}
    print(W, line["config"]["build_id"], {k[-44:]: (round(v.get("fp64_flop_per_launch", 0) / 1e12, 3), round(v.get("valu_busy", 0), 3),
                                                    round(v.get("valu_per_wave_abscissa", 0), 1)) for k, v in kern.items() if "integrate" in k})
try:        # where there is a git checkout (the build container, not the GPU box): the commit the evidence was collected at
    import subprocess
    prof["collected_at_git_head"] = subprocess.run(["git", "-C", R, "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
except Exception:
    pass
json.dump(prof, open(prof_path, "w"), indent=1, sort_keys=True)
for n in ("bench_default", "bench_faithful", "bench_c2pp", "bench_c3", "bench_c4", "bench_c5", "bench_c1", "bench_hstorage", "bench_mnm", "bench_gpus2_strong",
          "bench_gpus2_weak", "bench_nt128"):
    line = bench_line(os.path.join(G, n + ".log"))
    if line:
        json.dump(line, open(os.path.join(P, f"{tag}_{n}.json"), "w"))
        r = line["roofline"]
        print(n, round(line["value"]), "pt/s", "kernel_ms", round(r["kernel_ms"], 2), "x", r.get("kernel_launches_per_step"), "frac", r.get("frac"))
for f in ("parity_r03.json", "stages_r03.json"):
    if os.path.exists(os.path.join(G, f)):
        shutil.copy(os.path.join(G, f), os.path.join(P, f))
