#!/usr/bin/env python3
"""bench.py -- (t,r) drawdown points/s on the Neuman-1974 1024x256 sweep (BASELINE.json
configs[1], SURVEY.md section 8d "C2"): model 5 with beta = 0, fully penetrating well,
M = 26 (53 Laplace samples), tanh-sinh k = 6 / R = 4, 10 J0 intervals x 48 Gauss-Lobatto
nodes => 543 abscissae, 28 779 Laplace-Hankel samples per point, fp64 throughout.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode faithful|fast]

One "step" = one pass of the hot path over the whole synthetic sweep that is resident
in HBM (times tD[1024] with their split index, radii rD[256] in; h and dh per point out).  With N > 1 the
driver launches this file under torch.distributed.run; the flattened (t,r) index is
block-partitioned over ranks (weak scaling: every rank owns a full 1024x256 block of
a 1024 x 256N sweep), no data-path collective, and each step ends with the RCCL
all-gather of the (h, dh) results that the reference's single output file implies.

The JSON line carries
  roofline      : bound = fp64 VALU (this path is neither HBM- nor MFMA-bound);
                  achieved = algorithmic flop per launch / average kernel duration
                  (HIP events on the launch stream); peak = 78.6 TFLOP/s (fp64 vector,
                  = 1/2 of the 157.3 TF fp32 vector figure of MI355X_MICROARCH.md); the
                  measured fp64-FMA rate of this device and the HBM view are added.
  cpu_baseline  : the reference binary itself (oracle/_ref/O2/unconfined, flang -O2,
                  OpenMP on all host cores) when it was shipped with the repo, else the
                  C oracle; timed on a bounded sample of the same workload, rank 0, N=1.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- SURVEY.md section 8d flop convention (fp64, FMA = 2): reference formulation of a model-5 sample
FLOP_PER_SAMPLE = 1420.0
FLOP_TAIL_PER_POINT = 0.1e6
PEAK_FP64_VALU_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0


def c2_deck():
    from unconfined_amd.deck import Deck
    return Deck(quiet=0, model=5, dimless=False, timeseries=True, piezometer=True,
                Q=42.8, l=160.0, d=0.0, rw=0.3333, rc=0.3333, gammaSkin=1.0, timeType=1, timePar=[0.0, 1.0],
                b=160.0, Kr=0.24, kappa=0.4, Ss=2.9e-5, Sy=0.23, beta=0.0, MoenchM=1, MoenchAlpha=[-9999.0],
                ac=2.9, ak=0.37, psia=2.0, psik=0.22, usL=20.0, MNtype=2, order=5,
                M=26, alpha=1.0e-8, tol=1.0e-9, k=6, R=4, j0s=[1, 1], nacc=10, ord=50,
                tval=9999.9, rval=85.1, zTop=146.7, zBot=144.7, zOrd=2, rwobs=0.167, sF=1.0)


def workload_deck(name):
    """BASELINE.json configs; 'c2' is the headline and the default"""
    from unconfined_amd.deck import Deck
    c2 = c2_deck()
    if name == "c2":
        return c2, 1024, 256, "C2: Neuman-1974 (model 5, beta=0) fully penetrating"
    if name == "c2pp":     # same sweep, partially penetrating pumping well (cape-cod-neuman74.in geometry)
        return c2.replace(l=60.2, d=13.2), 1024, 256, "C2pp: Neuman-1974 (model 5, beta=0) partially penetrating"
    g = Deck.read(os.path.join(ROOT, "tests", "golden", "decks", {"c3": "c3_moench", "c4": "c4_malama_partpen",
                                                                "c5": "c5_mishra_fd64"}[name] + ".in"))
    if name == "c3":
        return g, 2048, 512, "C3: Moench 3-alpha delayed yield, screened observation well (nz=2)"
    if name == "c4":
        return g, 4096, 1024, "C4: Malama-2011 partial penetration (beta=2), k=7/R=5, nacc=12"
    return g, 1024, 256, "C5: Mishra-Neuman finite-difference vadose zone, 64 nodes"


def cpu_baseline(dk, ncores):
    """reference CPU throughput on a bounded sample (8 radii x 1024 times of the C2 sweep, ~10-15 s)"""
    from unconfined_amd.deck import TimeSpec
    ref = os.path.join(ROOT, "oracle", "_ref", "O2", "unconfined")
    radii = [16.0, 29.6, 54.9, 84.8, 160.0, 480.0, 960.0, 1600.0]      # spans the sweep's rD = 0.1 .. 10
    nt = 1024
    if os.path.exists(ref):
        work = tempfile.mkdtemp(prefix="ucf_cpu_")
        try:
            TimeSpec(True, -1, 8, nt).write(os.path.join(work, "time_c2.dat"))
            t0 = time.time()
            for i, r in enumerate(radii):
                d = dk.replace(rval=r, timeFileName="time_c2.dat", spaceFileName="unused.dat", outFileName=f"c2_{i}.out")
                d.write(os.path.join(work, f"c2_{i}.in"))
                env = dict(os.environ, OMP_NUM_THREADS=str(ncores))
                subprocess.run([ref, f"c2_{i}.in"], cwd=work, env=env, check=True, capture_output=True)
            dt = time.time() - t0
            vals = []
            for i in range(len(radii)):
                with open(os.path.join(work, f"c2_{i}.out"), errors="replace") as f:
                    vals += [[float(x) for x in ln.split()[:3]] for ln in f if ln.strip() and not ln.startswith("#")]
            rows = len(vals)
            if rows == nt * len(radii):
                return {"value": rows / dt, "unit": "points/s", "cores": ncores, "kind": "reference",
                        "sample": f"reference binary (flang -O2, OpenMP), C2 deck, {len(radii)} radii x {nt} times = {rows} points in {dt:.1f} s",
                        "_radii": radii, "_rows": np.array(vals)}
        except Exception as exc:  # fall through to the port
            print(f"[bench] reference binary unusable here ({exc}); timing the C oracle instead", file=sys.stderr)
        finally:
            shutil.rmtree(work, ignore_errors=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    from unconfined_amd.abi import params_from_deck
    O = Oracle()
    P = params_from_deck(dk)
    D = O.nondim(P)
    t = O.logspace(-1, 8, nt)
    tD = np.tile(t / D.Tc, len(radii))
    rD = np.repeat(np.array(radii) / D.Lc, nt)
    sv = np.ones(len(tD), np.int32)
    zD = np.array([145.7 / D.Lc])
    zl = O.zlay(D, zD)
    t0 = time.time()
    ho, dho = O.batch(P, tD, rD, sv, zD, zl, threads=ncores)
    dt = time.time() - t0
    return {"value": len(tD) / dt, "unit": "points/s", "cores": ncores, "kind": "port",
            "sample": f"C oracle (gcc -O2, OpenMP), C2 deck, {len(radii)} radii x {nt} times = {len(tD)} points in {dt:.1f} s",
            "_radii": radii, "_rows": np.stack([np.tile(t, len(radii)), ho[:, 0] * D.Hc, dho[:, 0] * D.Hc], axis=1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default=os.environ.get("UCF_BENCH_MODE", "fast"), choices=["faithful", "fast"])
    ap.add_argument("--workload", default="c2", choices=["c2", "c2pp", "c3", "c4", "c5"])
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--nr", type=int, default=0)
    ap.add_argument("--layout", default="auto", choices=["auto", "sample"])
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the drawdown path has no CPU fallback")
    # UCF_BENCH_BACKEND=gloo + UCF_BENCH_ONE_DEVICE=1: rehearsal of the N > 1 logic on a one-GPU box (all
    # ranks on cuda:0, gather staged through the host); the graded runs use RCCL, one rank per GPU
    backend = os.environ.get("UCF_BENCH_BACKEND", "nccl")
    if os.environ.get("UCF_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    from unconfined_amd import engine
    from unconfined_amd.abi import params_from_deck

    dk, nt_def, nr_def, wl_name = workload_deck(args.workload)
    args.nt = args.nt or nt_def
    args.nr = args.nr or nr_def
    P = params_from_deck(dk)
    plan = engine.Plan(P, mode=args.mode, layout=args.layout)
    D = plan.derived

    # ---- the sweep: 1024 log-spaced times x (256 * world) log-spaced radii, rank owns a block of 256 radii
    nt, nr = args.nt, args.nr
    t = engine.logspace(-1, 8, nt)
    tD = t / D.Tc
    sv_t = plan.split_vector(tD)
    rD_all = 10.0 ** engine.linspace(-1.0, 1.0, nr * world)
    rD_mine = rD_all[rank * nr:(rank + 1) * nr]
    npts = nt * nr
    zD = engine.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd) / D.Lc
    zl = plan.zlay(zD)
    nz = len(zD)

    dev = torch.device("cuda", local_rank)
    d_tD = torch.from_numpy(np.ascontiguousarray(tD)).to(dev)                 # [nt]
    d_rD = torch.from_numpy(np.ascontiguousarray(rD_mine)).to(dev)            # [nr]
    d_sv = torch.from_numpy(sv_t.astype(np.int32)).to(dev)                    # [nt]
    d_out = torch.zeros(2, npts * nz, dtype=torch.float64, device=dev)       # [h; dh]
    d_all = torch.zeros(world * 2, npts * nz, dtype=torch.float64, device=dev) if world > 1 else None
    stream = torch.cuda.current_stream()

    def gather():
        if backend == "nccl":
            dist.all_gather_into_tensor(d_all, d_out)
        else:                                   # rehearsal path only
            host_all = torch.empty(d_all.shape, dtype=d_all.dtype)
            dist.all_gather_into_tensor(host_all, d_out.cpu())
            d_all.copy_(host_all)

    def step():
        plan.drawdown_grid_device(nt, d_tD.data_ptr(), d_sv.data_ptr(), nr, d_rD.data_ptr(), zD, zl,
                                  d_out[0].data_ptr(), d_out[1].data_ptr(), stream=stream.cuda_stream)
        if world > 1:
            gather()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # dominant-kernel duration: HIP events recorded by the library on the launch stream right around that
    # kernel (ucf_plan_set_timing), read back after each of the same K timed steps; the events around the
    # whole call (all kernels of a step) are kept as `step_kernels_ms`
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    plan.set_timing(True)
    dom_ms, dom_name = [], ""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k > 0:
            try:
                ms_k, dom_name = plan.kernel_ms()       # waits for step k-1's kernel only
                dom_ms.append(ms_k)
            except Exception:
                pass
        ev[k][0].record(stream)
        plan.drawdown_grid_device(nt, d_tD.data_ptr(), d_sv.data_ptr(), nr, d_rD.data_ptr(), zD, zl,
                                  d_out[0].data_ptr(), d_out[1].data_ptr(), stream=stream.cuda_stream)
        ev[k][1].record(stream)
        if world > 1:
            gather()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    try:
        ms_k, dom_name = plan.kernel_ms()
        dom_ms.append(ms_k)
    except Exception:
        pass
    step_kernels_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    kern_ms = float(np.mean(dom_ms)) if dom_ms else step_kernels_ms
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # sanity: results are finite and the gathered copy equals the local one
    h_host = d_out[0].cpu().numpy()
    ok = bool(np.isfinite(h_host).all())
    if world > 1:
        ok = ok and bool(torch.equal(d_all[2 * rank:2 * rank + 2], d_out))

    if rank == 0:
        total_pts = npts * world * args.steps
        value = total_pts / elapsed
        samples_per_pt = D.nabs * D.np * nz
        per_sample = FLOP_PER_SAMPLE if dk.model in (3, 5) else (30.0 if dk.model == 0 else FLOP_PER_SAMPLE)
        if dk.model == 6 and dk.MNtype == 2:
            per_sample += dk.order * 68.0            # n*(1 cdiv + 3 cmul + 2 cadd)*2 passes
        flop_per_pt = per_sample * D.nabs * D.np + 0.35 * per_sample * D.nabs * D.np * (nz - 1) + FLOP_TAIL_PER_POINT * nz
        achieved_tf = flop_per_pt * npts / (kern_ms * 1e-3) * 1e-12
        try:
            fma_peak = engine.fp64_fma_peak()
        except Exception:
            fma_peak = None
        alg_bytes = (20 + 16 * nz) * npts
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):
            try:
                # measured on the full 1024 x 256 C2 sweep; traffic is proportional to the points of a launch
                traffic = json.load(open(tpath)).get(args.mode, {}).get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic = traffic * (npts / 262144.0) if wl_name.startswith("C2:") else None
            except Exception:
                traffic = None
        executed = None
        ppath = os.path.join(ROOT, "profiles", "pmc_r01.json")
        if os.path.exists(ppath) and wl_name.startswith("C2:") and args.mode == "fast":
            try:
                pm = json.load(open(ppath))["fast"]
                fl = pm.get("fp64_flop_executed_per_launch")
                if fl:
                    fl = fl * (npts / 262144.0)          # counted on the full 1024 x 256 sweep
                    executed = {"fp64_flop_per_launch": fl, "TFLOPs": fl / (kern_ms * 1e-3) * 1e-12,
                                "frac_of_peak": fl / (kern_ms * 1e-3) * 1e-12 / PEAK_FP64_VALU_TFLOPS,
                                "valu_busy": pm.get("valu_busy_per_simd"), "source": "profiles/pmc_r01.json (rocprofv3 --pmc of this workload)"}
            except Exception:
                executed = None
        line = {
            "metric": "(t,r) drawdown points/sec, Neuman-1974 1024x256 sweep; max |rel err| vs CPU ref",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{wl_name}, {nt} log-spaced times x {nr} "
                                   f"log-spaced radii per GPU, M={dk.M}, k={dk.k}/R={dk.R}, nacc={dk.nacc}, ord={dk.ord} ({D.nabs} abscissae, "
                                   f"{samples_per_pt} samples/point)",
                       "points_per_gpu": npts, "mode": args.mode, "layout": args.layout, "partition": "contiguous (t,r) blocks, one rank per GPU",
                       "results_finite_and_gather_consistent": ok},
            "roofline": {"bound": "fp64_valu", "achieved": achieved_tf, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / PEAK_FP64_VALU_TFLOPS, "traffic": traffic,
                         "kernel": dom_name or "all kernels of a step", "kernel_ms": kern_ms,
                         "step_kernels_ms": step_kernels_ms, "flop_per_point": flop_per_pt,
                         "convention": "SURVEY.md 8(d): 1.42 kflop per Laplace-Hankel sample as the reference formulates it (+0.1 Mflop tail per point); "
                                       "frac > 1 = the kernel needs fewer flops than that formulation, see `executed` for the instructions it really issues",
                         "executed": executed,
                         "peak_measured_fp64_fma": fma_peak,
                         "frac_of_measured_fma": (achieved_tf / fma_peak) if fma_peak else None,
                         "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                                 "achieved_GBs": alg_bytes / (kern_ms * 1e-3) * 1e-9, "peak_GBs": PEAK_HBM_GBS,
                                 "frac": alg_bytes / (kern_ms * 1e-3) * 1e-9 / PEAK_HBM_GBS}},
        }
        if world == 1 and not args.no_cpu:
            # the box shows every hardware thread of the host but a 1-GPU job owns a 16-core share; the
            # reference's OpenMP regions are 48-63 iterations long, so more threads only add overhead
            navail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            ncores = int(os.environ.get("UCF_CPU_THREADS", min(16, navail)))
            cb = cpu_baseline(dk, ncores)
            # the second half of the metric: max |rel err| of the GPU path against the CPU reference on the
            # sample the CPU leg just computed (same deck, same times; outside the timed region)
            try:
                rows = cb.pop("_rows")
                rr = np.array(cb.pop("_radii")) / D.Lc
                tDs = engine.logspace(-1, 8, 1024) / D.Tc
                hg, dg = plan.drawdown_grid(tDs, plan.split_vector(tDs), rr, zD, zl)
                hg = hg[:, :, 0].T.ravel() * D.Hc
                dg = dg[:, :, 0].T.ravel() * D.Hc
                eh = np.abs(hg - rows[:, 1]) / np.maximum(np.abs(rows[:, 1]), 1e-3)
                ed = np.abs(dg - rows[:, 2]) / np.maximum(np.abs(rows[:, 2]), 1e-3)
                line["accuracy_vs_cpu_ref"] = {
                    "points": int(len(eh)), "floor": 1e-3,
                    "h_max_rel": float(eh.max()), "h_median_rel": float(np.median(eh)), "h_frac_within_1e-10": float(np.mean(eh <= 1e-10)),
                    "dh_max_rel": float(ed.max()), "dh_median_rel": float(np.median(ed)), "dh_frac_within_1e-10": float(np.mean(ed <= 1e-10)),
                    "note": "reference -O2 vs -O3 -march=native builds differ by 1.7e-10 (h) / 4e-8 (dh) on this deck (SURVEY.md H1)"}
            except Exception as exc:
                line["accuracy_vs_cpu_ref"] = {"error": str(exc)}
            cb.pop("_rows", None); cb.pop("_radii", None)
            line["cpu_baseline"] = cb
            line["gpu_over_cpu"] = value / cb["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
