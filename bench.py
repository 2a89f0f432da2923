#!/usr/bin/env python3
"""bench.py -- (t,r) drawdown points/s on the Neuman-1974 1024x256 sweep (BASELINE.json
configs[1], SURVEY.md section 8d "C2"): model 5 with beta = 0, fully penetrating well,
M = 26 (53 Laplace samples), tanh-sinh k = 6 / R = 4, 10 J0 intervals x 48 Gauss-Lobatto
nodes => 543 abscissae, 28 779 Laplace-Hankel samples per point, fp64 throughout.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak] [--workload c2|c2pp|c3|c4|c5]
                  [--mode fast|faithful]

One "step" = one pass of the hot path over the whole synthetic sweep that is resident in HBM
(times tD[nt] with their split index, radii rD[nr] in; h and dh per point out).

N > 1: one process per GPU over RCCL.  Under torch.distributed.run (RANK / WORLD_SIZE in the
environment) this file is a rank; started plainly with --gpus N it launches its N ranks itself
(fresh child processes, before anything here touches a GPU) and fails if RCCL does not see N ranks.
  --scaling weak (default)    every rank owns a full nt x nr sweep (the metric's 1024 x 256 on C2) over every N-th radius
                              of an nt x nr*N sweep (cyclic: each rank spans rD = 0.1 ... 10 and costs the same): the points
                              are independent, per-GPU work is fixed as N grows.
  --scaling strong            the FIXED sweep of the workload (C2 1024x256, C4 4096x1024, ...) is cut into N contiguous
                              blocks of time rows (ucf_shard_rows: the reference's i loop, driver.f90:100); every
                              rank computes its rows in place in the full-size result arrays and one in-place
                              all-gather per array completes them on every rank (the reference's single output).
No data-path collective in either; the final gather is inside the timed step.  The other mode is measured after the
line of the timed region is complete and reported under "other_scaling" (under a time limit: the line is never lost to it).

The JSON line carries
  roofline      : bound = fp64 VALU (this path is neither HBM- nor MFMA-bound).  `achieved` = fp64 flop the
                  dominant kernel EXECUTES per launch / its average duration over the timed steps (HIP events recorded by
                  the library on the launch stream around every kernel); the executed-flop count comes from the
                  rocprofv3 --pmc profile under profiles/ that carries this library's build id (tools/gpu_final.sh
                  regenerates it in the same pass as the kernel trace) -- a profile of another build is refused and
                  the field is null.  `peak` = 78.6 TFLOP/s (fp64 vector = 1/2 of the 157.3 TF fp32 vector figure of
                  MI355X_MICROARCH.md).  `kernels` = the same for every kernel of the step.  The SURVEY 8(d)
                  convention figure (flop of the REFERENCE formulation / kernel time) is kept apart as
                  `time_to_solution_vs_reference_formulation`: it is not a utilisation.
  cpu_baseline  : the reference binary itself (oracle/_ref/O2/unconfined, flang -O2, OpenMP) when it was shipped with
                  the repo, else the C oracle; timed on a bounded sample of the same workload, rank 0, N = 1.
  accuracy_vs_cpu_ref : the second half of the metric, both flavours, on that sample (outside the timed region).
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- SURVEY.md section 8d flop convention (fp64, FMA = 2): reference formulation of a model-5 sample
FLOP_PER_SAMPLE = 1420.0
FLOP_TAIL_PER_POINT = 0.1e6
PEAK_FP64_VALU_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
PROFILE = os.path.join(ROOT, "profiles", "pmc_r03.json")


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
    g = Deck.read(os.path.join(ROOT, "tests", "golden", "decks", {"c3": "c3_moench", "c4": "c4_malama_partpen", "c5": "c5_mishra_fd64",
                                                                "c1": "c1_theis", "hstorage": "hstorage_partpen_lay2",
                                                                "mnm": "mishra_malama"}[name] + ".in"))
    if name == "c1":
        return g, 1024, 256, "C1: Theis (model 0), 1024 x 256 sweep"
    if name == "hstorage":
        return g, 1024, 256, "Hantush with wellbore storage (model 2), partially penetrating"
    if name == "mnm":
        return g, 1024, 256, "Mishra-Neuman, Malama's closed form (model 6 / MNtype 1)"
    if name == "c3":
        return g, 2048, 512, "C3: Moench 3-alpha delayed yield, screened observation well (nz=2)"
    if name == "c4":
        return g, 4096, 1024, "C4: Malama-2011 partial penetration (beta=2), k=7/R=5, nacc=12"
    return g, 1024, 256, "C5: Mishra-Neuman finite-difference vadose zone, 64 nodes"


def cpu_baseline(dk, ncores):
    """reference CPU throughput on a bounded sample (8 radii x 1024 times of the C2 sweep, ~10-15 s)"""
    import numpy as np
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default=os.environ.get("UCF_BENCH_MODE", "fast"), choices=["faithful", "fast"])
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"])
    ap.add_argument("--workload", default="c2", choices=["c2", "c2pp", "c3", "c4", "c5", "c1", "hstorage", "mnm"])
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--nr", type=int, default=0)
    ap.add_argument("--layout", default="auto", choices=["auto", "sample"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-other-scaling", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as fresh child processes (this
    process has not touched a GPU and never will) and pass on rank 0's line; non-zero exit if any rank fails"""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), UCF_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = time.time() + float(os.environ.get("UCF_BENCH_LAUNCH_TIMEOUT", "1500"))
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench] rank {r} exited with code {code}; stopping the others", file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
        if time.time() > deadline:
            print("[bench] ranks did not finish in time; stopping them", file=sys.stderr)
            for q in alive:
                procs[q].kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=30)
        except Exception:
            p.kill()
    return rc


def load_profile(build_id, workload, mode):
    """per-kernel counters of this workload from profiles/pmc_r03.json -- only if they were taken on THIS build"""
    try:
        prof = json.load(open(PROFILE))
    except Exception as exc:
        return None, f"no profile ({exc.__class__.__name__})"
    ent = prof.get("workloads", {}).get(workload, {}).get(mode)
    if not ent:
        return None, f"profile has no entry for workload {workload} / {mode}"
    if ent.get("build_id") != build_id:
        return None, f"profile was taken on build {ent.get('build_id')}, this library is build {build_id}: refused"
    return ent, f"profiles/pmc_r03.json (rocprofv3 --pmc, build {build_id}, {ent.get('points_per_launch')} points per launch)"


def worker(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for the wrong GPU count", file=sys.stderr)
        return 5
    # UCF_BENCH_BACKEND=gloo + UCF_BENCH_ONE_DEVICE=1: rehearsal of the N > 1 logic on a one-GPU box (all
    # ranks on cuda:0, gather staged through the host); the graded runs use RCCL, one rank per GPU
    backend = os.environ.get("UCF_BENCH_BACKEND", "nccl")
    one_device = os.environ.get("UCF_BENCH_ONE_DEVICE") == "1"
    ndev = torch.cuda.device_count()
    if ndev < 1 or (not one_device and local_rank >= ndev):
        print(f"[bench] rank {rank}: needs GPU {local_rank}, {ndev} visible (the drawdown path has no CPU fallback)", file=sys.stderr)
        return 3
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # how many ranks does the collective library really see?
        cnt = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(cnt)
        n_ranks = int(cnt.item())
        if n_ranks != args.gpus:
            if rank == 0:
                print(f"[bench] {backend} reports {n_ranks} ranks, --gpus asked for {args.gpus}", file=sys.stderr)
            return 4
    else:
        n_ranks = 1

    from unconfined_amd import engine, sharding
    from unconfined_amd.abi import params_from_deck

    dk, nt_def, nr_def, wl_name = workload_deck(args.workload)
    nt = args.nt or nt_def
    nr = args.nr or nr_def
    P = params_from_deck(dk)
    plan = engine.Plan(P, mode=args.mode, layout=args.layout)
    D = plan.derived
    t = engine.logspace(-1, 8, nt)
    tD = t / D.Tc
    sv_t = plan.split_vector(tD)
    zD = engine.linspace(dk.zBot, dk.zTop, 1 if dk.piezometer else dk.zOrd) / D.Lc
    zl = plan.zlay(zD)
    nz = len(zD)
    stream = torch.cuda.current_stream()
    d_tD = torch.from_numpy(np.ascontiguousarray(tD)).to(dev)                 # [nt]
    d_sv = torch.from_numpy(sv_t.astype(np.int32)).to(dev)                    # [nt]

    def to_host_gather(full_dev, gather_fn):           # rehearsal path only (gloo): stage through the host
        host = full_dev.cpu()
        gather_fn(host)
        full_dev.copy_(host)

    # ---- strong scaling: the fixed nt x nr sweep, rows block-partitioned (ucf_shard_rows), results in place
    row = nr * nz
    prow = sharding.padded_rows(nt, world)
    lo, hi = sharding.shard_rows(nt, world, rank)
    rD_fixed = 10.0 ** engine.linspace(-1.0, 1.0, nr)
    d_rD_fixed = torch.from_numpy(np.ascontiguousarray(rD_fixed)).to(dev)
    d_full = torch.zeros(2, prow * row, dtype=torch.float64, device=dev)      # [h; dh] x [prow][nr][nz]

    # the gather of the strong-scaling step belongs to the product: ucf_drawdown_grid_allgather issues the rank's rows and the
    # two in-place ncclAllGather calls on the same stream, over a communicator the library made itself (rank 0 draws the
    # id, torch.distributed only carries its 128 bytes to the other ranks).  UCF_BENCH_GATHER=torch: the all-gather of
    # torch.distributed instead (same layout; also what happens, loudly, if the library's communicator cannot be made)
    lib_comm = None
    gather_via = "none"

    def make_lib_comm():
        nonlocal lib_comm, gather_via
        if world == 1 or lib_comm:
            return
        gather_via = "torch.distributed all_gather_into_tensor" + ("" if backend == "nccl" else " (host-staged, rehearsal)")
        if backend == "nccl" and os.environ.get("UCF_BENCH_GATHER", "ucf") != "torch":
            def carry(buf):                       # the 128-byte id + validity byte, rank 0 -> all (every rank calls this)
                t = buf.to(dev)
                dist.broadcast(t, 0)
                torch.cuda.synchronize()
                return t.cpu()
            lib_comm, why = sharding.library_communicator(world, rank, dev, carry_id=carry,
                                                          timeout=float(os.environ.get("UCF_BENCH_COMM_TIMEOUT", "120")))
            if lib_comm:
                gather_via = "ucf_drawdown_grid_allgather (in-place ncclAllGather on the launch stream, communicator from ucf_comm_create)"
            else:
                print(f"[bench] rank {rank}: the library's RCCL communicator is unavailable ({why}); gathering with torch.distributed", file=sys.stderr)
                lib_comm = None
            ok_all = torch.tensor([1 if lib_comm else 0], device=dev)
            dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)          # all ranks or none
            if int(ok_all.item()) == 0 and lib_comm:
                engine.comm_destroy(lib_comm)
                lib_comm = None
                gather_via = "torch.distributed all_gather_into_tensor (a rank could not make the library's communicator)"

    def step_strong(gather=True):
        if lib_comm and gather:
            plan.drawdown_grid_allgather(lib_comm, rank, world, nt, d_tD.data_ptr(), d_sv.data_ptr(), nr, d_rD_fixed.data_ptr(), zD, zl,
                                         d_full[0].data_ptr(), d_full[1].data_ptr(), stream=stream.cuda_stream)
            return
        plan.drawdown_grid_shard_device(rank, world, nt, d_tD.data_ptr(), d_sv.data_ptr(), nr, d_rD_fixed.data_ptr(), zD, zl,
                                        d_full[0].data_ptr(), d_full[1].data_ptr(), stream=stream.cuda_stream)
        if world > 1 and gather:
            for a in (0, 1):
                if backend == "nccl":
                    sharding.allgather_rows_(d_full[a], nt, row, world, rank)
                else:
                    to_host_gather(d_full[a], lambda hst: sharding.allgather_rows_(hst, nt, row, world, rank))

    # ---- weak scaling: nt x (nr * world) sweep, every rank nr of its radii -- every world-th one (cyclic): each rank's sweep
    # spans the whole range rD = 0.1 ... 10 like the metric's own, so the ranks cost the same (contiguous blocks of radii do not:
    # the cosh form of the closure at large radii costs 1.3 x the exponential form at small ones, tools/shard_balance.py)
    rD_all = 10.0 ** engine.linspace(-1.0, 1.0, nr * world)
    d_rD_mine = torch.from_numpy(np.ascontiguousarray(rD_all[rank::world])).to(dev)
    d_out = d_all = None

    def alloc_weak():
        nonlocal d_out, d_all
        if d_out is None:
            d_out = torch.zeros(2, nt * row, dtype=torch.float64, device=dev)
            d_all = torch.zeros(world * 2, nt * row, dtype=torch.float64, device=dev) if world > 1 else None

    def step_weak(gather=True):
        plan.drawdown_grid_device(nt, d_tD.data_ptr(), d_sv.data_ptr(), nr, d_rD_mine.data_ptr(), zD, zl,
                                  d_out[0].data_ptr(), d_out[1].data_ptr(), stream=stream.cuda_stream)
        if world > 1 and gather:
            if backend == "nccl":
                dist.all_gather_into_tensor(d_all, d_out)
            else:
                host_all = torch.empty(d_all.shape, dtype=d_all.dtype)
                dist.all_gather_into_tensor(host_all, d_out.cpu())
                d_all.copy_(host_all)

    def timed(step_fn, steps, warmup, with_kernel_times):
        """the contract's timed region: W warm-up steps, barrier + synchronize, K steps, synchronize + barrier, max over ranks"""
        for _ in range(warmup):
            step_fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        plan.set_timing(with_kernel_times)
        per_kernel = {}
        order = []

        def collect():
            try:
                for name, ms, cnt in plan.kernel_times():   # waits for the previous step's kernels only
                    if name not in per_kernel:
                        per_kernel[name] = []
                        order.append(name)
                    per_kernel[name].append((ms, cnt))
            except Exception:
                pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            if k > 0 and with_kernel_times:
                collect()
            step_fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if with_kernel_times:
            collect()
        plan.set_timing(False)
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        # per kernel: average duration of ONE launch and launches per step (a step that walks the radii in chunks
        # launches every kernel once per chunk)
        return elapsed, [(n, float(np.sum([m for m, _ in per_kernel[n]]) / max(1, np.sum([c for _, c in per_kernel[n]]))),
                          int(round(np.mean([c for _, c in per_kernel[n]])))) for n in order]

    if args.scaling == "weak":
        alloc_weak()
    else:
        make_lib_comm()
    main_step = step_strong if args.scaling == "strong" else step_weak
    elapsed, kernels = timed(main_step, args.steps, args.warmup, True)
    pts_main = (nt * nr if args.scaling == "strong" else nt * nr * world)
    pts_launch = ((hi - lo) * nr if args.scaling == "strong" else nt * nr)       # points of one rank's launch

    # sanity: finite results, and every rank holds the same gathered sweep
    if args.scaling == "strong":
        hh = d_full[0][: nt * row]
        ok = bool(torch.isfinite(hh).all().item())
        if world > 1:
            chk = torch.stack([hh.sum(), d_full[1][: nt * row].sum()])
            if backend != "nccl":
                chk = chk.cpu()
            lo_, hi_ = chk.clone(), chk.clone()
            dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
            ok = ok and bool(torch.equal(lo_, hi_))
    else:
        ok = bool(torch.isfinite(d_out[0]).all().item())
        if world > 1:
            ok = ok and bool(torch.equal(d_all[2 * rank:2 * rank + 2], d_out))
    main_gather = gather_via if args.scaling == "strong" else ("torch.distributed all_gather_into_tensor" if world > 1 else "none")

    def other_scaling(line):
        """The other scaling mode, AFTER the line of the timed region is complete (rank 0 holds it): if this second
        measurement does not come back -- the library's own N-rank communicator runs here for the first time on a node
        this repository never had -- the line is printed without it instead of being lost with the run."""
        import threading

        def bail():
            if rank == 0:
                line["other_scaling"] = {"error": "did not finish within its time limit; the timed region above is unaffected"}
                print(json.dumps(line), flush=True)
            os._exit(0)
        dog = threading.Timer(float(os.environ.get("UCF_BENCH_OTHER_TIMEOUT", "240")), bail)
        dog.daemon = True
        dog.start()
        try:
            if args.scaling == "strong":
                alloc_weak()
                e2, _ = timed(step_weak, args.steps, 1, False)
                other = {"scaling": "weak", "value": nt * nr * world * args.steps / e2, "ms_per_step": e2 / args.steps * 1e3,
                         "points_per_step": nt * nr * world, "gather": "torch.distributed all_gather_into_tensor"}
            else:
                make_lib_comm()
                e2, _ = timed(step_strong, args.steps, 1, False)
                hh = d_full[0][: nt * row]
                chk = torch.stack([hh.sum(), d_full[1][: nt * row].sum()])
                if backend != "nccl":
                    chk = chk.cpu()
                lo_, hi_ = chk.clone(), chk.clone()
                dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
                dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
                other = {"scaling": "strong", "value": nt * nr * args.steps / e2, "ms_per_step": e2 / args.steps * 1e3,
                         "points_per_step": nt * nr, "gather": gather_via,
                         "results_finite_and_gather_consistent": bool(torch.isfinite(hh).all().item()) and bool(torch.equal(lo_, hi_))}
        except Exception as exc:
            other = {"error": f"{exc.__class__.__name__}: {exc}"}
        dog.cancel()
        if lib_comm:
            torch.cuda.synchronize()
            engine.comm_destroy(lib_comm)
        return other

    want_other = world > 1 and not args.no_other_scaling
    if rank != 0:
        if want_other:
            other_scaling(None)
        elif lib_comm:
            torch.cuda.synchronize()
            engine.comm_destroy(lib_comm)
        if world > 1:
            dist.destroy_process_group()
        return 0

    value = pts_main * args.steps / elapsed
    samples_per_pt = D.nabs * D.np * nz
    per_sample = FLOP_PER_SAMPLE if dk.model in (3, 5) else (30.0 if dk.model == 0 else FLOP_PER_SAMPLE)
    if dk.model == 6 and dk.MNtype == 2:
        per_sample += dk.order * 68.0            # n*(1 cdiv + 3 cmul + 2 cadd)*2 passes
    flop_per_pt = per_sample * D.nabs * D.np + 0.35 * per_sample * D.nabs * D.np * (nz - 1) + FLOP_TAIL_PER_POINT * nz
    try:
        fma_peak = engine.fp64_fma_peak()
    except Exception:
        fma_peak = None
    build_id = engine.build_id()
    prof, prof_src = load_profile(build_id, args.workload, args.mode)
    dom = max(kernels, key=lambda kv: kv[1] * kv[2]) if kernels else ("all kernels of a step", elapsed / args.steps * 1e3, 1)
    pts_kernel_launch = pts_launch / dom[2]                                # points one launch of the dominant kernel processes
    rows = []
    for name, ms, cnt in kernels:
        r = {"name": name, "ms": ms, "launches_per_step": cnt}
        pk = (prof or {}).get("kernels", {}).get(name)
        if pk:
            scale = (pts_launch / cnt) / float(prof["points_per_launch"])  # counters are proportional to the points of a launch
            if pk.get("fp64_flop_per_launch") is not None:
                fl = pk["fp64_flop_per_launch"] * scale
                r.update({"fp64_flop_per_launch": fl, "executed_TFLOPs": fl / (ms * 1e-3) * 1e-12,
                          "frac": fl / (ms * 1e-3) * 1e-12 / PEAK_FP64_VALU_TFLOPS})
            if pk.get("hbm_bytes_per_launch") is not None:
                by = pk["hbm_bytes_per_launch"] * scale
                r.update({"hbm_bytes_per_launch": by, "hbm_GBs": by / (ms * 1e-3) * 1e-9, "hbm_frac": by / (ms * 1e-3) * 1e-9 / PEAK_HBM_GBS})
            for k in ("valu_busy", "valu_per_wave_abscissa", "salu_per_wave_abscissa", "fp64_arith_per_wave_abscissa"):
                if pk.get(k) is not None:
                    r[k] = pk[k]
        rows.append(r)
    drow = next((r for r in rows if r["name"] == dom[0]), {})
    alg_bytes = (20 + 16 * nz) * pts_kernel_launch
    conv_tf = flop_per_pt * pts_kernel_launch / (dom[1] * 1e-3) * 1e-12
    line = {
        "metric": "(t,r) drawdown points/sec, Neuman-1974 1024x256 sweep; max |rel err| vs CPU ref",
        "value": value, "unit": "points/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{wl_name}, {nt} log-spaced times x {nr if args.scaling == 'strong' else nr * world} log-spaced radii"
                               f"{'' if args.scaling == 'strong' else f' ({nr} per GPU)'}, M={dk.M}, k={dk.k}/R={dk.R}, nacc={dk.nacc}, "
                               f"ord={dk.ord} ({D.nabs} abscissae, {samples_per_pt} samples/point)",
                   "points_per_step": pts_main, "points_per_gpu": pts_launch, "mode": args.mode, "layout": args.layout,
                   "partition": ("contiguous blocks of time rows (ucf_shard_rows), in-place all-gather of h and dh" if args.scaling == "strong"
                                 else "every world-th radius per rank (cyclic), all-gather of [h; dh]") + f", one rank per GPU, backend {backend if world > 1 else 'none'}",
                   "gather": main_gather,
                   "launcher": "self (bench.py spawned its ranks)" if os.environ.get("UCF_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if world > 1 else "single process"),
                   "build_id": build_id, "results_finite_and_gather_consistent": ok},
        "roofline": {"bound": "fp64_valu", "achieved": drow.get("executed_TFLOPs"), "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                     "frac": drow.get("frac"), "traffic": drow.get("hbm_bytes_per_launch"),
                     "kernel": dom[0], "kernel_ms": dom[1], "kernel_launches_per_step": dom[2], "points_per_kernel_launch": pts_kernel_launch,
                     "step_kernels_ms": float(sum(ms * c for _, ms, c in kernels)) if kernels else None,
                     "definition": "achieved = fp64 flop EXECUTED by the dominant kernel per launch (64 lanes x (2 FMA + ADD + MUL + TRANS) wave "
                                   "instructions, rocprofv3 --pmc) / its average launch duration over the timed steps (HIP events on the launch stream)",
                     "executed_flop_source": prof_src,
                     "kernels": rows,
                     "valu_busy": drow.get("valu_busy"),
                     "peak_measured_fp64_fma": fma_peak,
                     "frac_of_measured_fma": (drow["executed_TFLOPs"] / fma_peak) if (fma_peak and drow.get("executed_TFLOPs")) else None,
                     "time_to_solution_vs_reference_formulation": {
                         "flop_per_point": flop_per_pt, "TFLOPs_equivalent": conv_tf, "x_fp64_peak": conv_tf / PEAK_FP64_VALU_TFLOPS,
                         "convention": "SURVEY.md 8(d): 1.42 kflop per Laplace-Hankel sample as the REFERENCE formulates it (+0.1 Mflop tail per "
                                       "point) / dominant-kernel time; > 1 x peak means the kernel needs fewer flops than that formulation -- a "
                                       "time-to-solution figure, not a utilisation"},
                     "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                             "achieved_GBs": alg_bytes / (dom[1] * 1e-3) * 1e-9, "peak_GBs": PEAK_HBM_GBS,
                             "frac": alg_bytes / (dom[1] * 1e-3) * 1e-9 / PEAK_HBM_GBS}},
    }
    if want_other:
        line["other_scaling"] = other_scaling(line)
    elif lib_comm:
        torch.cuda.synchronize()
        engine.comm_destroy(lib_comm)
    if world == 1 and not args.no_cpu:
        # the box shows every hardware thread of the host but a 1-GPU job owns a 16-core share; the
        # reference's OpenMP regions are 48-63 iterations long, so more threads only add overhead
        navail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        ncores = int(os.environ.get("UCF_CPU_THREADS", min(16, navail)))
        cb = cpu_baseline(dk if args.workload == "c2" else c2_deck(), ncores)
        # the second half of the metric: max |rel err| of the GPU path against the CPU reference on the
        # sample the CPU leg just computed (same deck, same times; outside the timed region), both flavours
        try:
            rows_ref = cb.pop("_rows")
            rr = np.array(cb.pop("_radii"))
            acc = {"points": int(len(rows_ref)), "floor": 1e-3,
                   "note": "reference -O2 vs -O3 -march=native builds differ by 1.7e-10 (h) / 4e-8 (dh) on this deck (SURVEY.md H1)"}
            Pc = params_from_deck(c2_deck())
            for flavour in ("fast", "faithful"):
                pl2 = engine.Plan(Pc, mode=flavour)
                D2 = pl2.derived
                zD2 = np.array([145.7 / D2.Lc]); zl2 = pl2.zlay(zD2)
                tDs = engine.logspace(-1, 8, 1024) / D2.Tc
                hg, dg = pl2.drawdown_grid(tDs, pl2.split_vector(tDs), rr / D2.Lc, zD2, zl2)
                hg = hg[:, :, 0].T.ravel() * D2.Hc
                dg = dg[:, :, 0].T.ravel() * D2.Hc
                eh = np.abs(hg - rows_ref[:, 1]) / np.maximum(np.abs(rows_ref[:, 1]), 1e-3)
                ed = np.abs(dg - rows_ref[:, 2]) / np.maximum(np.abs(rows_ref[:, 2]), 1e-3)
                acc[flavour] = {
                    "h_max_rel": float(eh.max()), "h_median_rel": float(np.median(eh)), "h_frac_within_1e-10": float(np.mean(eh <= 1e-10)),
                    "dh_max_rel": float(ed.max()), "dh_median_rel": float(np.median(ed)), "dh_frac_within_1e-10": float(np.mean(ed <= 1e-10))}
                pl2.close()
            line["accuracy_vs_cpu_ref"] = acc
        except Exception as exc:
            line["accuracy_vs_cpu_ref"] = {"error": str(exc)}
        cb.pop("_rows", None); cb.pop("_radii", None)
        line["cpu_baseline"] = cb
        # the same sweep through the HOST-array entry of the boundary (ucf_drawdown_grid: numpy arrays in, h and dh copied back
        # over PCIe, the call returns when they are there) -- never `value`, reported beside it
        try:
            rD_h = np.ascontiguousarray(10.0 ** engine.linspace(-1.0, 1.0, nr))
            plan.drawdown_grid(tD, sv_t, rD_h, zD, zl)
            nh = 3
            t0 = time.perf_counter()
            for _ in range(nh):
                hh_, dd_ = plan.drawdown_grid(tD, sv_t, rD_h, zD, zl)
            eh_ = (time.perf_counter() - t0) / nh
            line["host_entry_pcie_inclusive"] = {"value": nt * nr / eh_, "unit": "points/s", "ms_per_step": eh_ * 1e3, "steps": nh,
                                                 "entry": "ucf_drawdown_grid (host arrays; inputs H2D, h and dh D2H, synchronous)",
                                                 "finite": bool(np.isfinite(hh_).all() and np.isfinite(dd_).all())}
        except Exception as exc:
            line["host_entry_pcie_inclusive"] = {"error": str(exc)}
        line["gpu_over_cpu"] = value / cb["value"] if args.workload == "c2" else None
    if world == 1 and not args.no_other_workloads and not args.no_cpu and args.workload == "c2" and not args.nt and not args.nr:
        # the other BASELINE.json configurations (and the models outside them) at FULL size on this GPU, after the timed
        # region: the same step (resident inputs, the grid entry on the current stream), 1 warm-up + 2 timed steps each,
        # per-kernel times from the library's event brackets
        others = {}
        for w in ("c2pp", "c3", "c4", "c5", "c1", "hstorage", "mnm"):
            try:
                dk2, nt2, nr2, nm2 = workload_deck(w)
                pl2 = engine.Plan(params_from_deck(dk2), mode=args.mode)
                D2 = pl2.derived
                tD2 = engine.logspace(-1, 8, nt2) / D2.Tc
                zD2 = engine.linspace(dk2.zBot, dk2.zTop, 1 if dk2.piezometer else dk2.zOrd) / D2.Lc
                zl2 = pl2.zlay(zD2)
                d_t2 = torch.from_numpy(np.ascontiguousarray(tD2)).to(dev)
                d_s2 = torch.from_numpy(pl2.split_vector(tD2).astype(np.int32)).to(dev)
                d_r2 = torch.from_numpy(np.ascontiguousarray(10.0 ** engine.linspace(-1.0, 1.0, nr2))).to(dev)
                d_o2 = torch.zeros(2, nt2 * nr2 * len(zD2), dtype=torch.float64, device=dev)

                def step2():
                    pl2.drawdown_grid_device(nt2, d_t2.data_ptr(), d_s2.data_ptr(), nr2, d_r2.data_ptr(), zD2, zl2, d_o2[0].data_ptr(), d_o2[1].data_ptr(),
                                             stream=stream.cuda_stream)
                step2()
                torch.cuda.synchronize()
                pl2.set_timing(True)
                nst = 2
                t0 = time.perf_counter()
                ks = {}
                for k in range(nst):
                    step2()
                    torch.cuda.synchronize()
                    for name, ms, cnt in pl2.kernel_times():
                        ks.setdefault(name, []).append((ms, cnt))
                el = time.perf_counter() - t0
                pl2.set_timing(False)
                others[w] = {"workload": f"{nm2}, {nt2} x {nr2}, nz={len(zD2)}", "value": nt2 * nr2 * nst / el, "unit": "points/s",
                             "ms_per_step": el / nst * 1e3, "steps": nst, "finite": bool(torch.isfinite(d_o2).all().item()),
                             "kernels": [{"name": n, "ms_per_launch": float(np.sum([m for m, _ in v]) / max(1, np.sum([c for _, c in v]))),
                                          "launches_per_step": int(round(np.mean([c for _, c in v])))} for n, v in ks.items()]}
                pl2.close()
                del d_o2
            except Exception as exc:
                others[w] = {"error": str(exc)}
        line["other_workloads"] = others
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
