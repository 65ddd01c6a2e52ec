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
An N > 1 run then also times BASELINE config 4 (2048 x 6144, the fixed domain split over the
N slabs: strong scaling) and reports it as the extra row `strong_scaling`; `--scaling strong`
makes that workload the headline of the line instead.

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
STRONG_GRID = (2048, 6144)  # BASELINE.json config 4: the fixed domain that N slabs share
# steps of the bench workload that the oracle repeats on the CPU (120: 20-25 s on 16 host threads; on the GPU ~47 ms
# queued ahead of the warm-up -- the clocks of a GPU that starts cold settle over ~50 ms, see ms_per_step_blocks)
PARITY_STEPS = int(os.environ.get("FCPT_BENCH_PARITY_STEPS", "120"))
# Settle protocol, the same for every leg of the line (headline, config table, strong-scaling row): before the W warm-up
# steps the leg's own context runs at least SETTLE_STEPS steps AND at least SETTLE_MS of GPU time, so that a short timed
# region measures the settled clocks (a GPU that starts cold ramps over ~50-100 ms) and any one-off work of
# fcpt_run_steps (the hipGraph capture of launch-bound grids) lies before it
SETTLE_STEPS = int(os.environ.get("FCPT_BENCH_SETTLE_STEPS", "150"))
SETTLE_MS = float(os.environ.get("FCPT_BENCH_SETTLE_MS", "100"))
PROFILE_STRIDE = 4  # timed region: every 4th launch of the dominant kernel carries the HIP-event pair
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# FP64 vector issue: 256 CUs x 4 SIMDs, one wave instruction per 4 cycles per SIMD for FP64 FMA/MUL/ADD
# (78.6 TFLOP/s = 1024 SIMDs x 16 lanes x 2 flop x 2.4 GHz), MI355X_MICROARCH.md
N_SIMD, CLOCK_GHZ, FP64_CYCLES_PER_WAVE_INSTR = 1024, 2.4, 4
VALU_PEAK_GWINST = N_SIMD * CLOCK_GHZ / FP64_CYCLES_PER_WAVE_INSTR  # 614.4 G wavefront instructions per second

# SURVEY.md section 8(d) / BASELINE.md section 3, the contract figure of the WHOLE step: 4-pass model
# A source+viscosity 5|7, B radial transport 8|10, C azimuthal transport 11|13, D velocities+floors+CFL 8|10
# => 32|40 doubles = 256|320 B per cell-update (isothermal | ideal EOS).  roofline.step_frac prices the step with it.
STEP_BYTES = (256, 320)
# ALGORITHMIC bytes of each kernel as built: the distinct doubles it must read + write per cell (isothermal | ideal).
# The fused kernels replace several passes of the model, so their own figure is far below the passes' sum; the
# roofline of a kernel is priced with ITS bytes (DESIGN.md section 4 derives each row).
OWN_DOUBLES = {
    "k_transport_fused": (6, 8),          # Sigma, v_r, v_phi(, e) in and out
    "k_transport_fused_therm": (6, 9),
    "k_transport_fused_wide": (6, 8),
    "k_transport_radial": (8, 10), "k_transport_theta_march": (9, 11),
    "k_source_march": (6, 6),             # Sigma, Phi, v_r, v_phi -> v_r', v_phi'
    "k_source_march_adi": (10, 10),       # Sigma, v_r, v_phi, e -> v_r', v_phi', e', Q+, Q-, Q+ - Q-
    "k_source_march_adi_wide": (10, 10),
    "k_src_fused": (7, 7), "k_av_fused": (5, 7), "k_visc_fused": (6, 7),
    "k_cfl_rings": (2, 6), "k_cfl_rings_bc": (2, 6), "k_cfl_cells": (4, 7), "k_pressure": (3, 2), "k_potential": (2, 2), "k_ring_mean": (1, 1),
}


def affinity_threads(cap=16):
    return max(1, min(cap, len(os.sched_getaffinity(0))))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (the headline): --nr rings per GPU; strong: BASELINE config 4, the 2048x6144 grid "
                         "split over the --gpus slabs (at N > 1 the weak line carries it as the row strong_scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the table of the other BASELINE configurations")
    ap.add_argument("--no-strong-row", action="store_true", help="N > 1: skip the strong-scaling row (config 4)")
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
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="host: REHEARSAL of the N-rank line on fewer GPUs than ranks -- the ranks share the visible GPU(s), the "
                         "process group is gloo and the slabs talk through the library's host-staged transport "
                         "(fcpt_comm_init_host); every code path of --gpus N except the RCCL transfers themselves")
    ap.add_argument("--rank-deadline", type=float, default=1500.0,
                    help="plain --gpus N > 1: seconds after which the parent ends ranks that are still running")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="no GPU work: start the ranks, rendezvous over gloo, all-reduce, print the rank bookkeeping "
                         "(the CPU test of the launcher half of --gpus N)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1,
                    help="with --dry-run-ranks: this rank exits with code 3 before the rendezvous (tests the parent's "
                         "handling of a dying rank)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# N ranks from a plain `python bench.py --gpus N`
def spawn_ranks(args) -> int:
    """Starts the N ranks as child processes and watches ALL of them: the first rank that exits non-zero (or the
    deadline) ends the others -- a rank that dies leaves its siblings blocked in a collective, and the parent never
    touches the GPU, so ending exactly the children it started and reporting their exit codes is all there is to do."""
    import torch  # device_count() does not initialise the GPU

    have = torch.cuda.device_count()
    if have < args.gpus and not args.dry_run_ranks and not (args.transport == "host" and have >= 1):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible: refusing to report a "
                         f"{have}-GPU number as a {args.gpus}-GPU one\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0 = open(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"bench_rank0_{os.getpid()}.out"), "w+b")
    for rank in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, affinity_threads(64) // args.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if rank == 0 else sys.stderr))
    deadline = time.monotonic() + args.rank_deadline
    rcs = [None] * len(procs)
    failed = None
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
                if rcs[i] not in (None, 0) and failed is None:
                    failed = f"rank {i} exited with code {rcs[i]}"
        if failed is None and time.monotonic() > deadline:
            failed = f"deadline of {args.rank_deadline:.0f} s passed"
        if failed is not None:
            for i, p in enumerate(procs):  # exactly the children this process started
                if rcs[i] is None:
                    p.terminate()
            t_kill = time.monotonic() + 10.0
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = p.wait(timeout=max(0.1, t_kill - time.monotonic()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[i] = p.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    text = out0.read().decode()
    out0.close()
    try:
        os.unlink(out0.name)
    except OSError:
        pass
    if failed is not None or any(rcs):
        sys.stderr.write(f"bench.py: {failed or 'a rank failed'}; rank exit codes {rcs}\n")
        return 1
    sys.stdout.write(text)
    sys.stdout.flush()
    return 0


# ---------------------------------------------------------------------------------------------------------------
class Leg:
    """One workload on this rank's GPU: the slab context, its communicator and the step loop (`run`)."""

    def __init__(self, env, d, *, bodies=None, slab=None, loopback=False, name=""):
        """env: the rank's libraries and process group; d: descriptor of the GLOBAL grid; slab = (rank, nranks)."""
        self.env, self.name = env, name
        lib, torch = env["lib"], env["torch"]
        from fargocpt_amd import binding as B, driver
        self.B = B
        self.d = d
        if slab is not None:
            d.rank, d.nranks = slab
        self.nslabs = d.nranks
        self.bodies = bodies
        self.radii = lib.radii(d)
        self.fields = lib.initial_fields(d.copy(), self.radii)  # slab-local
        self.ctx = driver.make_context(lib, d, fields=self.fields, radii=self.radii, bodies=bodies)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.adi = 1 if d.eos == B.EOS_IDEAL else 0
        self.multi = env["world"] > 1 and d.nranks > 1
        self.loopback = loopback
        self.torch_comm = None  # fallback: the transfers through torch.distributed (fargocpt_amd/parallel.py)
        self.comm_note = "none (one slab)"
        if self.multi or loopback:
            self._connect()

    def _connect(self):
        """The library's own RCCL communicator: slab 0 draws the id, torch.distributed's store carries it.  Rank 0
        ALWAYS broadcasts -- the id or the reason it has none -- so that no rank is left waiting in the broadcast;
        after it every rank knows whether all of them have a communicator (MAX all-reduce) and otherwise all of
        them take the torch.distributed transfers."""
        env, B = self.env, self.B
        lib, dist, torch, dev, rank = env["lib"], env["dist"], env["torch"], env["dev"], env["rank"]
        comm_error = ""
        if self.loopback:
            self.ctx.set_option("comm_loopback", 1)
            self.ctx.comm_init(lib.comm_unique_id())  # a failure here is the rehearsal's result: let it raise
            self.comm_note = "RCCL inside the library, one rank in loopback"
            return
        if env.get("transport") == "host":   # rehearsal: ranks that share a GPU
            env["links"] = env.get("links", 0) + 1
            box = [os.path.join("/dev/shm", f"fcpt_bench_{os.getpid()}_{env['links']}") if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            self.ctx.comm_init_host(box[0])
            self.comm_note = "host-staged transport of the library (fcpt_comm_init_host): REHEARSAL, ranks share a GPU"
            return
        box = [None, ""]
        if rank == 0:
            try:
                box[0] = lib.comm_unique_id()
            except B.FcptError as err:  # e.g. librccl not loadable from the library
                box[1] = str(err)
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            comm_error = f"rank 0 has no unique id: {box[1]}"
        else:
            try:
                self.ctx.comm_init(box[0])
            except B.FcptError as err:
                comm_error = str(err)
        # all ranks or none
        bad = torch.tensor([1.0 if comm_error else 0.0], dtype=torch.float64, device=env["tdev"])
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if bad.item() > 0:
            sys.stderr.write(f"bench.py: rank {rank}: RCCL inside the library unavailable ({comm_error or 'another rank'}); "
                             "transfers through torch.distributed instead\n")
            if not comm_error:
                self.ctx.comm_destroy()
            from fargocpt_amd.parallel import DistributedSlab
            self.torch_comm = DistributedSlab(self.ctx, device=dev)
            self.comm_note = "torch.distributed fallback (isend/irecv + all_reduce on RCCL's stream)"
        else:
            self.comm_note = ("RCCL inside the library: grouped ncclSend/ncclRecv of 7 ghost rings per neighbour + "
                              "ncclAllReduce(min) of dt, on the step's stream")

    def run(self, n):
        # dt stays on the device: CFL [-> MIN over the slabs] -> policy -> step [-> ghost exchange] -> post,
        # enqueued by the library on one stream, no host synchronisation inside
        if self.torch_comm is not None:
            for _ in range(n):
                self.torch_comm.step_async()
        else:
            self.ctx.run_steps(n, snap=False)

    def pre_loop(self):
        # main()'s and sim::init's pre-loop calls (main.cpp:117,147, simulation.cpp:462-474)
        ctx = self.ctx
        if self.torch_comm is not None:
            self.torch_comm.prepare()
        elif self.multi or self.loopback:
            ctx.calculate_timestep(ctx.cfl_allreduce())
            ctx.exchange()
            ctx.apply_boundary(0.0, False)
            ctx.calculate_timestep(ctx.cfl_allreduce())
            ctx.exchange()
        else:
            for _ in range(2):
                ctx.calculate_timestep(ctx.cfl())

    def sync(self):
        self.ctx.synchronize()
        self.env["torch"].cuda.synchronize()

    def settle(self, extra_steps=0):
        """>= SETTLE_STEPS (+ extra) steps and >= SETTLE_MS of GPU time on this context; returns the steps taken.
        Every rank runs the same count: the block length is fixed and the stop decision is all-reduced."""
        env = self.env
        done, t0 = 0, time.perf_counter()
        want = SETTLE_STEPS + extra_steps
        block = max(50, want)
        while True:
            self.run(block)
            self.sync()
            done += block
            more = 1.0 if (done < want or 1e3 * (time.perf_counter() - t0) < SETTLE_MS) and done < 200000 else 0.0
            if self.multi:
                t = env["torch"].tensor([more], dtype=env["torch"].float64, device=env["tdev"])
                env["dist"].all_reduce(t, op=env["dist"].ReduceOp.MAX)
                more = float(t.item())
            if not more:
                return done
            block = min(4 * block, 20000)

    def timed(self, steps):
        """`steps` steps between barrier + synchronize on both sides; (max over ranks, per-rank) seconds."""
        env = self.env
        torch, dist = env["torch"], env["dist"]
        if self.multi:
            dist.barrier()
        self.sync()
        t0 = time.perf_counter()
        self.run(steps)
        self.sync()
        if self.multi:
            dist.barrier()
        el = time.perf_counter() - t0
        per_rank = [el]
        if self.multi:
            t = torch.tensor([el], dtype=torch.float64, device=env["tdev"])
            gathered = [torch.zeros_like(t) for _ in range(env["world"])]
            dist.all_gather(gathered, t)
            per_rank = [float(g.item()) for g in gathered]
            el = max(per_rank)
        return el, per_rank

    def close(self):
        self.ctx.close()


def weak_desc(lib, setups, args, nslabs):
    d = setups.planet_disk(lib, args.nr * nslabs, args.nphi, adiabatic=args.eos == "ideal")
    if nslabs > 1:
        # weak scaling: keep dr/r of the 1-GPU grid, extend the disk outward
        d.rmax = d.rmin * (2.5 / 0.4) ** nslabs
        d.damping_time_radius_outer = d.rmax
    return d


def strong_desc(lib, setups, args):
    # BASELINE.json config 4: the 2048 x 6144 isothermal disk of config 2's physics, ONE domain for any number of slabs
    return setups.planet_disk(lib, STRONG_GRID[0], STRONG_GRID[1], adiabatic=args.eos == "ideal")


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

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run_ranks and rank == args.dry_run_fail_rank:
        sys.exit(3)

    import numpy as np
    import torch  # first: the HIP runtime torch bundles must be the one the library binds to
    import torch.distributed as dist

    import fargocpt_amd
    from fargocpt_amd import binding as B, driver, setups

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
    host_transport = args.transport == "host" and world > 1
    if local_rank >= torch.cuda.device_count() and not host_transport:
        raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    tdev = torch.device("cpu") if host_transport else dev   # where the tensors of torch.distributed's collectives live
    multi = world > 1
    rehearse = args.rehearse_exchange and not multi
    rccl_world = 1
    if multi:
        if host_transport:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        ones = torch.ones(1, dtype=torch.float64, device=tdev)
        dist.all_reduce(ones)  # what RCCL itself says the world is
        rccl_world = int(round(float(ones.item())))
        if rccl_world != args.gpus:
            raise SystemExit(f"bench.py: RCCL all-reduce over {rccl_world} ranks, --gpus {args.gpus}")
    if multi or rehearse:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))  # kernels and RCCL off the null stream

    lib = fargocpt_amd.load()
    env = {"lib": lib, "torch": torch, "dist": dist, "dev": dev, "tdev": tdev, "rank": rank, "world": world,
           "transport": "host" if host_transport else "rccl"}
    reh_rank, reh_n = (int(x) for x in args.rehearse_slab.split(":"))
    strong = args.scaling == "strong"
    if strong:
        d = strong_desc(lib, setups, args)
        slab = (rank, world)
    else:
        d = weak_desc(lib, setups, args, reh_n if rehearse else world)
        slab = (reh_rank, reh_n) if rehearse else (rank, world)
    leg = Leg(env, d, bodies=setups.jupiter_bodies(d), slab=slab, loopback=rehearse, name="headline")
    ctx, adi = leg.ctx, leg.adi
    leg.pre_loop()

    # ---- parity leg, device half: the same workload on a second context, PARITY_STEPS steps from the same initial
    # state; the oracle repeats them on the host cores after the timed region and the two end states are compared
    # (cpu_baseline.parity_max_rel).  Queued here, ahead of the settle steps.
    parity_ctx = None
    if world == 1 and not rehearse and not args.no_cpu_baseline:
        parity_ctx = driver.make_context(lib, d, fields=leg.fields, radii=leg.radii, bodies=leg.bodies)
        parity_ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            parity_ctx.calculate_timestep(parity_ctx.cfl())
        parity_ctx.run_steps(PARITY_STEPS)
    # ---- calibration: a per-kernel pass (HIP events around every launch of three steps) finds the dominant kernel.
    # It comes BEFORE the settle steps: fcpt_profile_stop synchronises and creates events, a pause of milliseconds on
    # the host during which the GPU idles and its clocks sag -- with the calibration between the settle steps and the
    # timed region (round 2) a 20-step region read 3 % above the blocks that followed it.  The events of the timed
    # region are created here too.
    cal = 3
    ctx.profile_start(None, max_launches=64 * cal)
    leg.run(cal)
    prof = ctx.profile_stop()
    dominant = max(prof, key=lambda k: prof[k][0])
    names = lib.kernel_names()
    ctx.profile_start([names.index(dominant)], max_launches=args.steps + 8)
    ctx.profile_stop()
    settle_steps = leg.settle() + cal
    leg.run(args.warmup)

    # ---- timed region ---------------------------------------------------------
    # HIP events around the dominant kernel, live in the timed region (roofline.kernel_ms): an event pair costs ~3 us of
    # stream time, 1.6 % of a step if every launch carries one -- every PROFILE_STRIDE-th launch does
    stride = PROFILE_STRIDE if args.steps >= 4 * PROFILE_STRIDE else 1
    ctx.set_option("profile_stride", stride)
    ctx.profile_start([names.index(dominant)], max_launches=args.steps + 8)
    elapsed, per_rank_s = leg.timed(args.steps)
    dom = ctx.profile_stop()[dominant]
    ctx.set_option("profile_stride", 1)
    per_rank_ms = [1e3 * t / args.steps for t in per_rank_s]

    # the same block again, a few times: a timed region of K x 0.4 ms can end before the GPU clocks have settled
    blocks = []
    for _ in range(max(0, args.settle_blocks)):
        leg.sync()
        b0 = time.perf_counter()
        leg.run(args.steps)
        leg.sync()
        blocks.append(1e3 * (time.perf_counter() - b0) / args.steps)

    st = ctx.state()
    finite = all(np.isfinite(v).all() for v in st.values())
    exchange_ok = None
    if multi:
        f = torch.tensor([1.0 if finite else 0.0], dtype=torch.float64, device=tdev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        finite = bool(f.item() > 0)
        exchange_ok = check_exchange(st, ctx, dist, torch, rank, world, tdev)
    slab_nr = ctx.nr
    comm_note = leg.comm_note
    leg.close()

    # ---- N > 1: BASELINE config 4 as the strong-scaling row (every rank takes part) ----------------------------------
    strong_row = None
    if multi and not strong and not args.no_strong_row:
        strong_row = strong_scaling_row(env, args, setups, np)

    if rank == 0:
        nr_total, nphi = d.nr_global, d.nphi
        cells = (args.nr * args.nphi) if rehearse else nr_total * nphi  # (rehearsal: the one slab that ran)
        value = cells * args.steps / elapsed
        dom_ms = dom[0] / max(1, dom[1])
        slab_cells = slab_nr * nphi
        grid_key = f"{args.nr}x{args.nphi}" if not strong else f"{STRONG_GRID[0]}x{STRONG_GRID[1]}"
        pmc = load_pmc(grid_key, "ideal" if adi else "isothermal") if world == 1 and not rehearse else {}
        physics = ("ideal-gas" if adi else "locally-isothermal") + " disk + 1 Jupiter-mass planet (examples/config.yml physics: " \
            "alpha=1e-3, TW artificial viscosity, reflecting BC + damping, FARGO transport, Euler)"
        if strong:
            workload = f"{nr_total}x{nphi} {physics}, BASELINE config 4: one domain split over {world} radial slab(s)"
        else:
            workload = f"{nr_total}x{nphi} {physics}, {args.nr} rings per GPU"
        out = {
            "metric": "cell-updates/s on Nr x Nphi polar grid", "value": value, "unit": "cell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "grid": [nr_total, nphi], "parallelism": f"radial slabs x{world}",
                       "finite": bool(finite),
                       "rehearsal": (f"slab {reh_rank} of {reh_n}" if rehearse else
                                     (f"{world} ranks on {torch.cuda.device_count()} GPU(s), host-staged transport: NOT a scaling "
                                      "number" if host_transport else False)),
                       # N > 1: after the last step every slab's ghost rings equal its neighbours' rows [7,14) /
                       # [nr-14,nr-7) bit for bit, and all slabs hold the same clock (the MIN-reduced dt)
                       "ghost_rings_and_clock_consistent": exchange_ok,
                       "communication": comm_note},
            "rccl_world": rccl_world,
            "ms_per_step_per_rank": per_rank_ms,
            # what ran untimed on this context before the timed region, and the same K-step block timed again
            # right after it (clock ramp / settling)
            "untimed_steps_before_timed_region": args.warmup + settle_steps,
            "settle_protocol": f">= {SETTLE_STEPS} steps and >= {SETTLE_MS:.0f} ms of this workload on the timed context before "
                               "the warm-up; the same for every row of `configs` and for `strong_scaling`",
            "untimed_other": "2 CFL + CalculateTimeStep calls of sim::init; three calibration steps with HIP-event pairs "
                             "around every kernel (they find the dominant kernel) ahead of the settle steps"
                             + (f"; before the settle steps, {PARITY_STEPS} steps of the same workload on a second context "
                                "(device half of cpu_baseline's parity check), queued on the same stream"
                                if parity_ctx is not None else ""),
            "ms_per_step_blocks": blocks,
            "roofline": roofline(dominant, dom_ms, dom[1], stride, adi, slab_cells, pmc, value, world),
            # rocprofv3 names (fcpt_kernel_name): comparable with profiles/*_kernel_stats.csv line by line
            "kernel_ms_per_step": {k: v[0] / cal for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]},
        }
        if strong_row is not None:
            out["strong_scaling"] = strong_row
        if world == 1 and not rehearse and not args.no_configs:
            out["configs"] = config_table(env, args, setups)
        if parity_ctx is not None:
            hip_state = parity_ctx.state()
            parity_ctx.close()
            out["cpu_baseline"] = cpu_baseline(d, leg.fields, leg.radii, leg.bodies, hip_state)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def roofline(dominant, dom_ms, launches, stride, adi, slab_cells, pmc, value, world):
    """The dominant kernel against the resource that binds it.  `frac` is always a fraction of the resource `bound`
    names and cannot exceed 1:
      * bound "valu_fp64" (the fused marching kernels: the PMC profile of this workload, profiles/pmc_latest*.json,
        shows far fewer HBM bytes than the vector pipeline could wait for): achieved = VALU wavefront instructions
        per launch (SQ_INSTS_VALU) / kernel time, peak = 1024 SIMDs x 2.4 GHz / 4 cycles per FP64 instruction;
      * bound "hbm" otherwise (and whenever no PMC profile of this workload is committed): achieved = the kernel's
        ALGORITHMIC bytes (OWN_DOUBLES x 8 B x the cells of one launch) / kernel time, peak = 8 TB/s.
    The HBM view is reported beside it in either case (`hbm`), with the PMC traffic (`traffic`), and `step_frac`
    prices the whole step with the contract's 256 | 320 B per cell-update."""
    t = dom_ms * 1e-3
    own_bytes = OWN_DOUBLES.get(dominant, (0, 0))[adi] * 8 * slab_cells
    traffic = pmc.get("hbm_bytes_per_launch", {}).get(dominant)
    valu_busy = pmc.get("valu_busy", {}).get(dominant)
    wave_insts = pmc.get("valu_insts_per_launch", {}).get(dominant)
    hbm_achieved = own_bytes / t / 1e9 if t > 0 else 0.0
    hbm = {"algorithmic_bytes_per_launch": own_bytes,
           "algorithmic_doubles_per_cell": OWN_DOUBLES.get(dominant, (0, 0))[adi],
           "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
           "traffic_frac": (traffic / t / 1e9 / HBM_PEAK_GBS) if traffic and t > 0 else None}
    issue = wave_insts / t / 1e9 if wave_insts and t > 0 else None  # G wavefront instructions per second
    valu_bound = issue is not None and traffic is not None and issue / VALU_PEAK_GWINST > (traffic / t / 1e9) / HBM_PEAK_GBS
    if valu_bound:
        head = {"bound": "valu_fp64", "achieved": issue, "peak": VALU_PEAK_GWINST,
                "unit": "G wavefront-instr/s (FP64 rate: 4 cycles each)", "frac": issue / VALU_PEAK_GWINST}
    else:
        head = {"bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": hbm_achieved / HBM_PEAK_GBS}
    head.update({
        "kernel": dominant, "traffic": traffic, "kernel_ms": dom_ms, "launches": launches,
        "launches_timed_every": stride, "hbm": hbm,
        "valu_busy": valu_busy, "valu_wave_insts_per_launch": wave_insts,
        # the whole step against SURVEY.md 8(d)'s 256|320 B per cell-update
        "step_frac": value * STEP_BYTES[adi] / (HBM_PEAK_GBS * 1e9 * world),
        "pmc_source": pmc.get("source"),
        "note": "frac = fraction of the resource `bound` names (never > 1). valu_fp64: SQ_INSTS_VALU per launch (PMC "
                "profile of this workload) x 4 cycles / (1024 SIMDs x 2.4 GHz x the HIP-event kernel time measured "
                "here); hbm: the kernel's own algorithmic bytes / kernel time / 8 TB/s, PMC bytes in `traffic`. "
                "step_frac is the contract number of BASELINE.md section 3 for the whole step."})
    return head


def strong_scaling_row(env, args, setups, np):
    """BASELINE config 4 on the ranks of this run: 2048 x 6144, one domain, `world` radial slabs of 2048/world (+ 14
    overlap) rings -- same settle / warm-up / barrier protocol as the headline, value = cells of the whole domain /
    max-over-ranks time."""
    lib, rank, world = env["lib"], env["rank"], env["world"]
    d = strong_desc(lib, setups, args)
    leg = Leg(env, d, bodies=setups.jupiter_bodies(d), slab=(rank, world), name="strong")
    leg.pre_loop()
    settled = leg.settle()
    leg.run(args.warmup)
    elapsed, per_rank = leg.timed(args.steps)
    st = leg.ctx.state()
    ok = check_exchange(st, leg.ctx, env["dist"], env["torch"], rank, world, env["tdev"])
    finite = all(np.isfinite(v).all() for v in st.values())
    nr_local = leg.ctx.nr
    note = leg.comm_note
    leg.close()
    cells = d.nr_global * d.nphi
    return {"scaling": "strong", "workload": f"BASELINE config 4: {d.nr_global}x{d.nphi} isothermal disk + Jupiter, one domain "
                                             f"over {world} radial slabs ({nr_local} rings on rank 0 incl. overlap)",
            "grid": [d.nr_global, d.nphi], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "untimed_steps_before_timed_region": settled + args.warmup,
            "value": cells * args.steps / elapsed, "unit": "cell-updates/s", "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_per_rank": [1e3 * t / args.steps for t in per_rank],
            "step_frac_per_gpu": cells * args.steps / elapsed * STEP_BYTES[leg.adi] / (HBM_PEAK_GBS * 1e9 * world),
            "ghost_rings_and_clock_consistent": ok, "finite_on_rank0": bool(finite), "communication": note}


def check_exchange(st, ctx, dist, torch, rank, world, dev):
    """What CommunicateBoundaries + the MIN all-reduce must leave behind, checked across the ranks: ghost rows
    [0,7) equal the inner neighbour's rows [nr-14,nr-7), ghost rows [nr-7,nr) the outer neighbour's rows [7,14)
    (the boundary conditions of the post step only touch the first and last slab's outermost rings -- except the
    reflecting v_r condition, which the reference applies without a rank guard, reflecting.cpp:15-40, and this library
    with it: v_r rows 0, 1 and nr-1 of EVERY slab are rewritten after the exchange and are left out of the comparison),
    and every slab's clock shows the same time and dt."""
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
        expect.append((buf, rows(0), (0, 1)))
    if rank < world - 1:
        buf = torch.empty_like(rows(0))
        ops += [dist.P2POp(dist.isend, rows(nr - 2 * G), rank + 1), dist.P2POp(dist.irecv, buf, rank + 1)]
        expect.append((buf, rows(nr - G), (G - 1,)))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    iv = names.index("vrad")
    for got, mine, skip in expect:
        for g in skip:   # v_r ghost rows the unguarded reflecting condition rewrites
            got[iv, g] = 0.0
            mine[iv, g] = 0.0
        same = bool(torch.equal(got, mine))
        if not same and os.environ.get("FCPT_BENCH_DEBUG"):
            diff = (got - mine).abs()
            sys.stderr.write(f"check_exchange rank {rank}: ghost rows differ, per field max |diff| "
                             f"{[float(diff[q].max()) for q in range(diff.shape[0])]}, per ghost row of field 0 "
                             f"{[float(diff[0, g].max()) for g in range(G)]}\n")
        ok = ok and same
    clk = ctx.clock
    if os.environ.get("FCPT_BENCH_DEBUG"):
        sys.stderr.write(f"check_exchange rank {rank}: time {clk.time!r} last_dt {clk.last_dt!r} rows ok {ok}\n")
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


def config_table(env, args, setups):
    """Step times of the other BASELINE.json configurations on this GPU, each on the headline's protocol: its own
    context, sim::init's pre-loop calls, the settle steps (>= SETTLE_STEPS steps and >= SETTLE_MS of GPU time), W
    warm-up steps, then `steps` steps timed between synchronisations (device-resident dt loop), with the whole step
    priced against the 256 | 320 B model.  Configs 1 and 5 are narrow grids: launch-bound, the fraction says so;
    `graph_replays` counts the hipGraphLaunch calls the library issued INSIDE the timed region (`graph_cycle` steps each;
    0 = plain launches)."""
    lib = env["lib"]
    from fargocpt_amd import binding as B

    rows = []

    def one(name, d, bodies=None, steps=None):
        leg = Leg(env, d, bodies=bodies, name=name)
        leg.pre_loop()
        settled = leg.settle()
        n = steps or args.steps
        leg.run(max(3, args.warmup))
        r0 = leg.ctx.get_option("graph_replays")
        el, _ = leg.timed(n)
        replays = leg.ctx.get_option("graph_replays") - r0
        ms = 1e3 * el / n
        cells = d.nr_global * d.nphi
        rows.append({"config": name, "grid": [d.nr_global, d.nphi], "steps": n, "ms_per_step": ms,
                     "untimed_steps_before_timed_region": settled + max(3, args.warmup),
                     "cell_updates_per_s": cells / (ms * 1e-3),
                     "step_frac": cells / (ms * 1e-3) * STEP_BYTES[leg.adi] / (HBM_PEAK_GBS * 1e9),
                     "graph_replays": replays, "graph_cycle": leg.ctx.get_option("graph_cycle"),
                     "coop_step": leg.ctx.get_option("coop_active")})
        leg.close()

    d = setups.spreading_ring(lib, 128, 384)
    one("1: spreading ring 128x384, isothermal, constant nu", d, steps=max(args.steps, 1000))
    d = setups.planet_disk(lib, 512, 1536)
    one("2: isothermal disk + Jupiter 512x1536", d, setups.jupiter_bodies(d), steps=max(args.steps, 200))
    d = setups.planet_disk(lib, 1024, 3072, adiabatic=True)
    one("3: ideal EOS + alpha viscosity + viscous heating 1024x3072", d, setups.jupiter_bodies(d), steps=max(args.steps, 100))
    d = setups.planet_disk(lib, 2048, 4096, adiabatic=True)
    one("3 at the headline grid: ideal EOS 2048x4096", d, setups.jupiter_bodies(d))
    d = setups.planet_disk(lib, 2048, 6144)
    one("4 on one GPU: isothermal 2048x6144", d, setups.jupiter_bodies(d))
    d = setups.shocktube(lib, 4096, 4, "SN")
    one("5: shock tube 4096x4, SN artificial viscosity", d, steps=max(args.steps, 1000))
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
