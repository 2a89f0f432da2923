"""copy the evidence of tools/gpu_final.sh from gpurun_out/ (scratch) into profiles/ (tracked)"""
import collections, csv, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
ks = sorted(glob.glob(os.path.join(G, "final_trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], os.path.join(P, f"{tag}_kernel_stats.csv"))
for n in ("bench_default", "bench_faithful"):
    src = os.path.join(G, n + ".log")
    if os.path.exists(src):
        line = open(src).read().strip().splitlines()[-1]
        open(os.path.join(P, f"{tag}_{n}.json"), "w").write(line + "\n")
def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    files = sorted(glob.glob(os.path.join(G, d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:                       # newest run only (gpurun merges, it does not clean)
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()}
fetch, write, sq, f64 = counters("final_FETCH_SIZE"), counters("final_WRITE_SIZE"), counters("final_sq"), counters("final_f64")
kernels = sorted(set(fetch) | set(write))
tot_f = sum(fetch.get(k, {}).get("FETCH_SIZE", 0.0) for k in kernels if "ucf_" in k)
tot_w = sum(write.get(k, {}).get("WRITE_SIZE", 0.0) for k in kernels if "ucf_" in k)
dom = [k for k in kernels if "integrate_kernel" in k] or [k for k in kernels if "point_kernel" in k]
dom_f = fetch.get(dom[0], {}).get("FETCH_SIZE", 0.0) if dom else 0.0
dom_w = write.get(dom[0], {}).get("WRITE_SIZE", 0.0) if dom else 0.0
traffic = {"fast": {
    "per_kernel_KB_per_launch": {k: {"FETCH_SIZE": fetch.get(k, {}).get("FETCH_SIZE"), "WRITE_SIZE": write.get(k, {}).get("WRITE_SIZE")} for k in kernels if "ucf_" in k},
    "dominant_kernel": dom[0] if dom else None,
    "hbm_bytes_per_launch": (2 * dom_f + dom_w) * 1024,
    "hbm_bytes_per_step_raw": (tot_f + tot_w) * 1024, "hbm_bytes_per_step": (2 * tot_f + tot_w) * 1024,
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python3 bench.py --steps 2 --warmup 1 --no-cpu` (C2, fast flavour; "
            "one step = abscissa_kernel + integrate_kernel<2,1> + finish_kernel<1,64> + point_kernel<2,1> (unfinished items only) + dehoog_tiles_kernel; "
            "averages per dispatch, KB). bytes = 2*FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: FETCH_SIZE reports half of wide coalesced reads on "
            "gfx950; narrow/scalar reads are uncalibrated, so this is an upper bound). hbm_bytes_per_launch is the dominant kernel alone, "
            "hbm_bytes_per_step the sum over the step. Algorithmic bytes per step: 9.4 MB. The rest is deliberate: integrate_kernel hands the "
            "(R+1+nacc) accumulators of every work item to finish_kernel through HBM (15 KB x 217 088 items = 3.3 GB written once, read once: "
            "~1 % of HBM peak at this rate) so that the abscissa loop runs at 4 waves/SIMD instead of 2, and the transform totlap makes one round "
            "trip (222 MB) between the lane = time layout and the lane = Laplace-index inversion."}}
json.dump(traffic, open(os.path.join(P, "traffic_r01.json"), "w"), indent=1)
pk = [k for k in sq if "integrate_kernel" in k] or [k for k in sq if "point_kernel" in k]
if pk:
    c = dict(sq[pk[0]])
    c.update(f64.get(pk[0], {}))
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    flop = None
    if "SQ_INSTS_VALU_FMA_F64" in c:
        flop = 64.0 * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_TRANS_F64"])
    json.dump({"fast": {"kernel": pk[0], "per_launch": c, "kernel_cycles": cyc,
                        "valu_busy_per_simd": c["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024),
                        "valu_instructions_per_wave_abscissa": c["SQ_INSTS_VALU"] / c["SQ_WAVES"] / 543,
                        "salu_instructions_per_wave_abscissa": c["SQ_INSTS_SALU"] / c["SQ_WAVES"] / 543,
                        "fp64_flop_executed_per_launch": flop,
                        "note": "rocprofv3 --pmc (two passes) over python3 bench.py --steps 2 --warmup 1 --no-cpu; averages per dispatch; SQ_* cycle counters are "
                                "quad-cycles, GRBM_GUI_ACTIVE is summed over 8 XCDs; fp64_flop_executed = 64 lanes x (2 FMA + ADD + MUL + TRANS) wave instructions"}},
              open(os.path.join(P, "pmc_r01.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in traffic["fast"].items() if k != "note"}, indent=1))
