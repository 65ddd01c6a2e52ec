#!/usr/bin/env python3
"""Headline benchmark: cell-updates/s of the per-timestep gas update on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full hydro step (CFL reduction, source + viscosity substeps, boundary
rings, FARGO transport, ghost exchange, derived quantities) of the 2048 x 4096
locally-isothermal disk + Jupiter-mass planet (examples/config.yml physics, BASELINE.json
config 2 at the grid the metric is quoted on).  With N > 1 every rank owns 2048 rings
(weak scaling; the log grid is extended outward so dr/r stays constant), neighbours
exchange 7 ghost rings per step and the CFL step is MIN-reduced over the slabs -- both
inside the library over RCCL (fcpt_comm_init / fcpt_run_steps), one process per GPU.

Ranks: under `python -m torch.distributed.run` the ranks are the launcher's (RANK,
LOCAL_RANK, WORLD_SIZE, MASTER_* from the environment).  Started plainly with --gpus N > 1,
this process starts the N ranks itself as child processes, before anything here touches a
GPU; with fewer than N GPUs visible it exits non-zero instead of reporting a smaller run.

Prints one JSON line on rank 0 (see the driver contract).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NR_PER_GPU, NPHI = 2048, 4096
# steps of the bench workload that the oracle repeats on the CPU (120: 20-25 s on 16 host threads; on the GPU ~47 ms
# queued ahead of the warm-up -- the clocks of a GPU that starts cold settle over ~50 ms, see ms_per_step_blocks)
PARITY_STEPS = int(os.environ.get("FCPT_BENCH_PARITY_STEPS", "120"))
SETTLE_STEPS = int(os.environ.get("FCPT_BENCH_SETTLE_STEPS", "150"))  # untimed, ahead of the warm-up (clock ramp)
PROFILE_STRIDE = 4  # timed region: every 4th launch of the dominant kernel carries the HIP-event pair
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# FP64 vector issue: 256 CUs x 4 SIMDs, one wave instruction per 4 cycles per SIMD for FP64 FMA/MUL/ADD
# (78.6 TFLOP/s = 1024 SIMDs x 16 lanes x 2 flop x 2.4 GHz), MI355X_MICROARCH.md
N_SIMD, CLOCK_GHZ, FP64_CYCLES_PER_WAVE_INSTR = 1024, 2.4, 4

# Algorithmic (minimum distinct read + written) doubles per cell of each kernel, isothermal /
# adiabatic; derivation in DESIGN.md section "Kernels".  SURVEY.md section 8(d)'s pass model:
# A source+viscosity 5|7, B radial transport 8|10, C azimuthal transport 11|13,
# D velocities+floors+CFL 8|10 => 32|40 doubles = 256|320 B per cell-update.
STEP_BYTES = (256, 320)
# share of SURVEY.md 8(d)'s pass model (doubles per cell, isothermal | adiabatic) that a fused kernel stands for:
# this is the ALGORITHMIC figure `roofline.achieved` is priced with
MODEL_PASSES = {
    "k_transport_fused": ("B+C+D", (27, 33)),
    "k_transport_theta_march": ("C+D", (19, 23)),
    "k_transport_radial": ("B", (8, 10)),
    "k_source_march": ("A", (5, 7)),
}
# the kernel's own minimal traffic (distinct doubles it must read + write per cell)
OWN_DOUBLES = {
    "k_transport_radial": (8, 10), "k_transport_theta_march": (9, 11), "k_transport_fused": (6, 8),
    "k_source_march": (6, 10), "k_src_fused": (7, 7), "k_av_fused": (5, 7), "k_visc_fused": (6, 7),
    "k_cfl_cells": (4, 7), "k_pressure": (3, 2), "k_potential": (2, 2), "k_ring_mean": (1, 1),
}


def affinity_threads(cap=16):
    return max(1, min(cap, len(os.sched_getaffinity(0))))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the table of the other BASELINE configurations")
    ap.add_argument("--nr", type=int, default=NR_PER_GPU, help="rings per GPU")
    ap.add_argument("--nphi", type=int, default=NPHI)
    ap.add_argument("--eos", choices=["isothermal", "ideal"], default="isothermal",
                    help="ideal: BASELINE config 3 physics (energy equation, viscous heating) on the same grid")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="one GPU: a middle slab that sends its ghost rings to itself through a 1-rank RCCL "
                         "communicator (what the exchange + MIN all-reduce add to a step; not a headline number)")
    ap.add_argument("--rehearse-slab", default="1:3", metavar="R:N",
                    help="with --rehearse-exchange: which slab of how many (the weak-scaling geometry of N GPUs; an inner "
                         "or outer slab talks to itself on its one neighbour side only)")
    ap.add_argument("--settle-blocks", type=int, default=4,
                    help="after the timed region: this many more blocks of --steps steps, timed one by one "
                         "(reported as ms_per_step_blocks: shows a clock ramp over a short timed region)")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="no GPU work: start the ranks, rendezvous over gloo, all-reduce, print the rank bookkeeping "
                         "(the CPU test of the launcher half of --gpus N)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# N ranks from a plain `python bench.py --gpus N`
def spawn_ranks(args) -> int:
    import torch  # device_count() does not initialise the GPU

    have = torch.cuda.device_count()
    if have < args.gpus and not args.dry_run_ranks:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible: refusing to report a "
                         f"{have}-GPU number as a {args.gpus}-GPU one\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, affinity_threads(64) // args.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:
            p.kill()  # exactly the child this process started
            rcs.append(-9)
    if any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        return 1
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return 0


# ---------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    # the contract is ONE JSON line on stdout: libraries that print there (RCCL's version banner at
    # communicator creation) are sent to stderr for the whole run, the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("RCCL_LOG_LEVEL", "0")
    os.environ.setdefault("OMP_NUM_THREADS", str(affinity_threads()))
    os.environ.setdefault("GOMP_SPINCOUNT", "100000")  # idle OpenMP threads spin ~0.1 ms, then sleep

    import numpy as np
    import torch  # first: the HIP runtime torch bundles must be the one the library binds to
    import torch.distributed as dist

    import fargocpt_amd
    from fargocpt_amd import binding as B, driver, setups

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run_ranks:
        dist.init_process_group("gloo")
        ones = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(ones)
        dist.barrier()
        if rank == 0:
            os.write(json_fd, (json.dumps({"dry_run": True, "n_gpus": world, "rccl_world": int(ones.item()),
                                           "config": {"parallelism": f"radial slabs x{world}"}}) + "\n").encode())
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1
    rehearse = args.rehearse_exchange and not multi
    rccl_world = 1
    if multi:
        dist.init_process_group("nccl", device_id=dev)
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones)  # what RCCL itself says the world is
        rccl_world = int(round(float(ones.item())))
        if rccl_world != args.gpus:
            raise SystemExit(f"bench.py: RCCL all-reduce over {rccl_world} ranks, --gpus {args.gpus}")
    if multi or rehearse:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))  # kernels and RCCL off the null stream

    lib = fargocpt_amd.load()
    reh_rank, reh_n = (int(x) for x in args.rehearse_slab.split(":"))
    nslabs = reh_n if rehearse else world
    nr_global = args.nr * nslabs
    d = setups.planet_disk(lib, nr_global, args.nphi, adiabatic=args.eos == "ideal")
    if nslabs > 1:
        # weak scaling: keep dr/r of the 1-GPU grid, extend the disk outward
        d.rmax = d.rmin * (2.5 / 0.4) ** nslabs
        d.damping_time_radius_outer = d.rmax
    d.rank, d.nranks = (reh_rank, reh_n) if rehearse else (rank, world)
    bodies = setups.jupiter_bodies(d)

    radii = lib.radii(d)
    fields = lib.initial_fields(d.copy(), radii)  # slab-local
    ctx = driver.make_context(lib, d, fields=fields, radii=radii, bodies=bodies)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    adi = 1 if d.eos == B.EOS_IDEAL else 0

    torch_comm = None  # fallback: the transfers through torch.distributed (fargocpt_amd/parallel.py)
    if multi or rehearse:
        # the library's own RCCL communicator: slab 0 draws the id, the torch.distributed store carries it
        comm_error = ""
        try:
            if rehearse:
                ctx.set_option("comm_loopback", 1)
                uid = lib.comm_unique_id()
            else:
                box = [lib.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                uid = box[0]
            ctx.comm_init(uid)
        except B.FcptError as err:   # e.g. librccl not loadable from the library: say so and keep the run alive
            comm_error = str(err)
        if multi:
            # all ranks or none: a rank that could not create its communicator sends every rank to the fallback
            bad = torch.tensor([1.0 if comm_error else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if bad.item() > 0:
                sys.stderr.write(f"bench.py: rank {rank}: RCCL inside the library unavailable ({comm_error or 'another rank'}); "
                                 "transfers through torch.distributed instead\n")
                if not comm_error:
                    ctx.comm_destroy()
                from fargocpt_amd.parallel import DistributedSlab
                torch_comm = DistributedSlab(ctx, device=dev)
        elif comm_error:
            raise SystemExit(f"bench.py: {comm_error}")

    # ---- one step ------------------------------------------------------------
    def run(n):
        # dt stays on the device: CFL [-> MIN over the slabs] -> policy -> step [-> ghost exchange] -> post,
        # enqueued by the library on one stream, no host synchronisation inside
        if torch_comm is not None:
            for _ in range(n):
                torch_comm.step_async()
        else:
            ctx.run_steps(n, snap=False)

    def pre_loop():
        # main()'s and sim::init's pre-loop calls (main.cpp:117,147, simulation.cpp:462-474)
        if torch_comm is not None:
            torch_comm.prepare()
        elif multi or rehearse:
            ctx.calculate_timestep(ctx.cfl_allreduce())
            ctx.exchange()
            ctx.apply_boundary(0.0, False)
            ctx.calculate_timestep(ctx.cfl_allreduce())
            ctx.exchange()
        else:
            for _ in range(2):
                ctx.calculate_timestep(ctx.cfl())

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    pre_loop()

    # ---- parity leg, device half: the same workload on a second context, PARITY_STEPS steps from the same initial
    # state; the oracle repeats them on the host cores after the timed region and the two end states are compared
    # (cpu_baseline.parity_max_rel).  Queued here, ahead of the warm-up: the timed region then starts on a GPU
    # that has been busy for ~50 ms instead of ~2 ms (see ms_per_step_blocks for what that is worth).
    parity_ctx = None
    if world == 1 and not rehearse and not args.no_cpu_baseline:
        parity_ctx = driver.make_context(lib, d, fields=fields, radii=radii, bodies=bodies)
        parity_ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            parity_ctx.calculate_timestep(parity_ctx.cfl())
        parity_ctx.run_steps(PARITY_STEPS)
    # The GPU's clocks settle over ~100 ms of load (ms_per_step_blocks of a 20-step run: 0.382 timed, 0.375, 0.371, 0.372
    # after it with the parity leg alone ahead of it): SETTLE_STEPS further steps of this workload on the timed context,
    # so that a short timed region measures the settled rate.
    settle_steps = SETTLE_STEPS
    if multi or rehearse:
        # no parity leg here (the oracle is a one-slab checker): the same number of steps on the timed context itself,
        # so that N > 1 and N = 1 start their timed regions on equally settled clocks
        settle_steps += PARITY_STEPS
    run(settle_steps)

    # ---- warm-up, with a per-kernel calibration pass to find the dominant kernel --
    cal = min(3, max(1, args.warmup))
    ctx.profile_start(None, max_launches=64 * cal)
    run(cal)
    prof = ctx.profile_stop()
    dominant = max(prof, key=lambda k: prof[k][0])
    names = lib.kernel_names()
    if args.warmup > cal:
        run(args.warmup - cal)

    # ---- timed region ---------------------------------------------------------
    # HIP events around the dominant kernel, live in the timed region (roofline.kernel_ms): an event pair costs ~3 us of
    # stream time, 1.6 % of a step if every launch carries one -- every PROFILE_STRIDE-th launch does
    stride = PROFILE_STRIDE if args.steps >= 4 * PROFILE_STRIDE else 1
    ctx.set_option("profile_stride", stride)
    ctx.profile_start([names.index(dominant)], max_launches=args.steps + 8)
    if multi:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    if multi:
        dist.barrier()
    t1 = time.perf_counter()
    dom = ctx.profile_stop()[dominant]
    ctx.set_option("profile_stride", 1)
    elapsed_rank = t1 - t0
    elapsed = elapsed_rank
    per_rank_ms = [1e3 * elapsed_rank / args.steps]
    if multi:
        t = torch.tensor([elapsed_rank], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank_ms = [1e3 * float(g.item()) / args.steps for g in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same block again, a few times: a timed region of K x 0.4 ms can end before the GPU clocks have settled
    blocks = []
    for _ in range(max(0, args.settle_blocks)):
        sync()
        b0 = time.perf_counter()
        run(args.steps)
        sync()
        blocks.append(1e3 * (time.perf_counter() - b0) / args.steps)

    st = ctx.state()
    finite = all(np.isfinite(v).all() for v in st.values())
    exchange_ok = None
    if multi:
        f = torch.tensor([1.0 if finite else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        finite = bool(f.item() > 0)
        exchange_ok = check_exchange(st, ctx, dist, torch, rank, world, dev)

    if rank == 0:
        cells = args.nr * world * args.nphi  # (rehearsal: the one slab that ran)
        value = cells * args.steps / elapsed
        dom_ms = dom[0] / max(1, dom[1])
        slab_cells = ctx.nr * args.nphi
        pmc = load_pmc(f"{args.nr}x{args.nphi}", "ideal" if adi else "isothermal")
        traffic = pmc.get("hbm_bytes_per_launch", {}).get(dominant)
        valu_busy = pmc.get("valu_busy", {}).get(dominant)
        wave_insts = pmc.get("valu_insts_per_launch", {}).get(dominant)
        # ALGORITHMIC bytes per launch: SURVEY.md 8(d)'s per-cell figure of the passes this kernel stands for
        model = MODEL_PASSES.get(dominant)
        doubles = model[1][adi] if model else OWN_DOUBLES.get(dominant, (0, 0))[adi]
        algo_bytes = doubles * 8 * slab_cells
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        own_bytes = OWN_DOUBLES.get(dominant, (0, 0))[adi] * 8 * slab_cells
        issue_frac = (wave_insts * FP64_CYCLES_PER_WAVE_INSTR / (N_SIMD * CLOCK_GHZ * 1e9 * dom_ms * 1e-3)
                      if wave_insts and dom_ms > 0 else None)
        # what binds the kernel: the algorithmic-bytes fraction says how close the launch is to the time the
        # 4-pass model needs at 8 TB/s; the kernel itself moves `own` bytes and is limited by FP64 vector issue
        bound = "valu_fp64" if (valu_busy or 0) >= 0.6 and traffic and traffic < 0.5 * algo_bytes else "hbm"
        out = {
            "metric": "cell-updates/s on Nr x Nphi polar grid", "value": value, "unit": "cell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.nr * world}x{args.nphi} {'ideal-gas' if adi else 'locally-isothermal'} disk + 1 Jupiter-mass planet "
                                   "(examples/config.yml physics: alpha=1e-3, TW artificial viscosity, "
                                   "reflecting BC + damping, FARGO transport, Euler), "
                                   f"{args.nr} rings per GPU",
                       "grid": [args.nr * world, args.nphi], "parallelism": f"radial slabs x{world}",
                       "finite": bool(finite), "rehearsal": (f"slab {reh_rank} of {reh_n}" if rehearse else False),
                       # N > 1: after the last step every slab's ghost rings equal its neighbours' rows [7,14) /
                       # [nr-14,nr-7) bit for bit, and all slabs hold the same clock (the MIN-reduced dt)
                       "ghost_rings_and_clock_consistent": exchange_ok,
                       "communication": ("torch.distributed fallback (isend/irecv + all_reduce on RCCL's stream)"
                                         if torch_comm is not None else
                                         "RCCL inside the library: grouped ncclSend/ncclRecv of 7 ghost rings per "
                                         "neighbour + ncclAllReduce(min) of dt, on the step's stream")
                       if (multi or rehearse) else "none (one slab)"},
            "rccl_world": rccl_world,
            "ms_per_step_per_rank": per_rank_ms,
            # what ran untimed on this context before the timed region, and the same K-step block timed again
            # right after it (clock ramp / settling)
            "untimed_steps_before_timed_region": args.warmup + settle_steps,
            "untimed_other": "2 CFL + CalculateTimeStep calls of sim::init; the first min(3, W) warm-up steps carry "
                             "HIP-event pairs around every kernel (calibration of the dominant kernel)"
                             + (f"; before the warm-up, {PARITY_STEPS} steps of the same workload on a second context "
                                "(device half of cpu_baseline's parity check), queued on the same stream"
                                if parity_ctx is not None else "")
                             + (f"; before the warm-up, {settle_steps} further steps of this workload on the timed context "
                                "(settled GPU clocks; N > 1 adds the parity leg's share, which it does not run)"
                                if settle_steps else ""),
            "ms_per_step_blocks": blocks,
            "roofline": {"bound": bound, "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": dom_ms, "launches": dom[1], "launches_timed_every": stride,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "algorithmic_passes": model[0] if model else "own traffic",
                         # the whole step against SURVEY.md 8(d)'s 256|320 B per cell-update
                         "step_frac": value * STEP_BYTES[adi] / (HBM_PEAK_GBS * 1e9 * world),
                         # the kernel's own minimal traffic (it fuses the model's passes) and its HBM fraction
                         "own": {"bytes_per_launch": own_bytes,
                                 "hbm_frac": own_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if dom_ms > 0 else None},
                         # FP64 vector pipeline: busy fraction (PMC) and issue-rate fraction
                         # = wave instructions x 4 cycles / (1024 SIMDs x 2.4 GHz x kernel time)
                         "valu_busy": valu_busy, "valu_wave_insts_per_launch": wave_insts,
                         "valu_issue_frac": issue_frac,
                         "note": "frac prices the launch with the algorithmic bytes of the passes of SURVEY.md 8(d) it "
                                 "replaces; the fused kernel moves far less (own.bytes_per_launch, traffic = PMC) and is "
                                 "bound by FP64 vector issue (valu_busy, valu_issue_frac), not by HBM"},
            "kernel_ms_per_step": {k: v[0] / cal for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]},
        }
        ctx.close()
        if world == 1 and not rehearse and not args.no_configs:
            out["configs"] = config_table(lib, args)
        if parity_ctx is not None:
            hip_state = parity_ctx.state()
            parity_ctx.close()
            out["cpu_baseline"] = cpu_baseline(d, fields, radii, bodies, hip_state)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    else:
        ctx.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def check_exchange(st, ctx, dist, torch, rank, world, dev):
    """What CommunicateBoundaries + the MIN all-reduce must leave behind, checked across the ranks: ghost rows
    [0,7) equal the inner neighbour's rows [nr-14,nr-7), ghost rows [nr-7,nr) the outer neighbour's rows [7,14)
    (the boundary conditions of the post step only touch the first and last slab's outermost rings), and every
    slab's clock shows the same time and dt."""
    import numpy as np
    G = 7
    names = [k for k in ("sigma", "vrad", "vazi", "energy") if k in st]
    nr = st["sigma"].shape[0]

    def rows(lo):
        return torch.from_numpy(np.stack([st[k][lo:lo + G] for k in names])).to(dev)

    ok = True
    ops, expect = [], []
    if rank > 0:
        buf = torch.empty_like(rows(0))
        ops += [dist.P2POp(dist.isend, rows(G), rank - 1), dist.P2POp(dist.irecv, buf, rank - 1)]
        expect.append((buf, rows(0)))
    if rank < world - 1:
        buf = torch.empty_like(rows(0))
        ops += [dist.P2POp(dist.isend, rows(nr - 2 * G), rank + 1), dist.P2POp(dist.irecv, buf, rank + 1)]
        expect.append((buf, rows(nr - G)))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    for got, mine in expect:
        ok = ok and bool(torch.equal(got, mine))
    clk = ctx.clock
    t = torch.tensor([clk.time, -clk.time, clk.last_dt, -clk.last_dt, 1.0 if ok else 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    t = t.cpu().numpy()
    return bool(t[0] == -t[1] and t[2] == -t[3] and t[4] > 0)


def load_pmc(workload, eos):
    """PMC counters of the committed profile run of the same workload (profiles/run_pmc.sh)."""
    for name in ("pmc_latest.json", f"pmc_latest_{eos}.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            try:
                rec = json.load(open(p))
                if rec.get("workload") == workload and rec.get("eos", "isothermal") == eos:
                    return rec
            except Exception:
                pass
    return {}


def config_table(lib, args):
    """Driver-run step times of the other BASELINE.json configurations on this GPU (ms per step over
    `steps` steps after `warmup`, device-resident dt loop), with the whole step priced against the 256 | 320 B
    model.  Configs 1 and 5 are narrow grids: launch-bound, the fraction says so."""
    import torch
    from fargocpt_amd import binding as B, driver, setups

    rows = []

    def one(name, d, bodies=None, steps=None):
        radii = lib.radii(d)
        fields = lib.initial_fields(d.copy(), radii)
        ctx = driver.make_context(lib, d, fields=fields, radii=radii, bodies=bodies)
        for _ in range(2):
            ctx.calculate_timestep(ctx.cfl())
        n = steps or args.steps
        ctx.run_steps(max(3, args.warmup))
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.run_steps(n)
        ctx.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        cells = d.nr_global * d.nphi
        adi = 1 if d.eos == B.EOS_IDEAL else 0
        rows.append({"config": name, "grid": [d.nr_global, d.nphi], "steps": n, "ms_per_step": ms,
                     "cell_updates_per_s": cells / (ms * 1e-3),
                     "step_frac": cells / (ms * 1e-3) * STEP_BYTES[adi] / (HBM_PEAK_GBS * 1e9),
                     "graph": ctx.get_option("graph_steps")})
        ctx.close()

    d = setups.spreading_ring(lib, 128, 384)
    one("1: spreading ring 128x384, isothermal, constant nu", d, steps=max(args.steps, 200))
    d = setups.planet_disk(lib, 512, 1536)
    one("2: isothermal disk + Jupiter 512x1536", d, setups.jupiter_bodies(d))
    d = setups.planet_disk(lib, 1024, 3072, adiabatic=True)
    one("3: ideal EOS + alpha viscosity + viscous heating 1024x3072", d, setups.jupiter_bodies(d))
    d = setups.planet_disk(lib, 2048, 4096, adiabatic=True)
    one("3 at the headline grid: ideal EOS 2048x4096", d, setups.jupiter_bodies(d))
    d = setups.planet_disk(lib, 2048, 6144)
    one("4 on one GPU: isothermal 2048x6144", d, setups.jupiter_bodies(d))
    d = setups.shocktube(lib, 4096, 4, "SN")
    one("5: shock tube 4096x4, SN artificial viscosity", d, steps=max(args.steps, 200))
    d = setups.planet_disk(lib, 2048, 4096)
    d.stabilize_viscosity = 1
    one("headline workload with StabilizeViscosity: 1 (pseudo-implicit viscous update in the marching kernel)", d,
        setups.jupiter_bodies(d))
    return rows


def cpu_baseline(d, fields, radii, bodies, hip_state):
    """The CPU oracle (oracle/fargo_oracle.c, a C+OpenMP restatement of the reference loops) timed on this
    box's host cores on a bounded sample of the same workload -- and used as the checker of the bench
    workload itself: `hip_state` is what the HIP path made of the same initial state in the same
    PARITY_STEPS steps; the two end states are compared (parity_max_rel, the bar is 1e-10)."""
    import ctypes
    import numpy as np
    from fargocpt_amd import binding as B, driver

    so = os.path.join(ROOT, "oracle", "libfargo_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    orc = B.Library(ctypes.CDLL(so), "orc_")
    threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    ctx = driver.make_context(orc, d, fields=fields, radii=radii, bodies=bodies)
    for _ in range(2):
        ctx.calculate_timestep(ctx.cfl())
    t0 = time.perf_counter()
    n = PARITY_STEPS
    el_first = None
    for k in range(n):
        ctx.run_steps(1)
        if el_first is None:
            el_first = time.perf_counter() - t0  # the first step pages the grids in: not part of the rate
    el = time.perf_counter() - t0
    cells = d.nr_global * d.nphi
    rate = cells * (n - 1) / (el - el_first)
    ref_state = ctx.state()
    ctx.close()
    parity = {}
    for k, b in ref_state.items():
        a = hip_state[k]
        parity[k] = float(np.abs(a - b).max() / np.abs(b).max())

    out = {"value": rate, "unit": "cell-updates/s", "cores": threads, "kind": "port",
           "sample": f"{n} steps of the same {d.nr_global}x{d.nphi} workload (the first one untimed), "
                     f"oracle/fargo_oracle.c (-O2, OpenMP, {threads} threads)",
           "parity_steps": n, "parity_max_rel": parity, "parity_ok": bool(max(parity.values()) <= 1e-10)}
    cal = os.path.join(ROOT, "profiles", "cpu_calibration.json")
    if os.path.exists(cal):
        try:
            out["vs_reference"] = json.load(open(cal))
        except Exception:
            pass
    return out


if __name__ == "__main__":
    main()
