#!/usr/bin/env python3
"""Headline benchmark: cell-updates/s of the per-timestep gas update on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full hydro step (CFL reduction, source + viscosity substeps, boundary
rings, FARGO transport, ghost exchange, derived quantities) of the 2048 x 4096
locally-isothermal disk + Jupiter-mass planet (examples/config.yml physics, BASELINE.json
config 2 at the grid the metric is quoted on).  With N > 1 every rank owns 2048 rings
(weak scaling; the log grid is extended outward so dr/r stays constant) and neighbours
exchange 7 ghost rings per step over RCCL.

Prints one JSON line on rank 0 (see the driver contract).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NR_PER_GPU, NPHI = 2048, 4096
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# Algorithmic (minimum distinct read + written) doubles per cell of each kernel, isothermal /
# adiabatic; derivation in DESIGN.md section "Kernels".  SURVEY.md section 8(d)'s pass model:
# A source+viscosity 5|7, B radial transport 8|10, C azimuthal transport 11|13,
# D velocities+floors+CFL 8|10 => 32|40 doubles = 256|320 B per cell-update.
ALGO_DOUBLES = {
    "k_transport_radial": (8, 10),   # read Sigma,vr,vphi(,e) -> write rm+-,L+-,Sigma(,e)
    "k_transport_theta1": (11, 13),  # read 5(6) + vphi -> write 5(6)
    "k_transport_theta2": (10, 12),  # read 5(6) -> write 5(6) (shifted)
    "k_velocities": (8, 10),         # read 5(6) -> write vr,vphi,Sigma(,e)
    "k_source_march": (6, 6),            # read Sigma,Phi,vr,vphi -> write vr,vphi (isothermal one-pass source step)
    "k_transport_theta_march": (9, 11),   # read 5(6) + vphi -> write vr,vphi,Sigma(,e): passes C and D of the model
    "k_transport_fused": (6, 8),          # read Sigma,vr,vphi(,e) -> write Sigma,vr,vphi(,e): passes B+C+D in one kernel
    "k_src_fused": (7, 7), "k_av_fused": (5, 7), "k_visc_fused": (6, 7),
    "k_source_vr": (6, 6), "k_source_va": (4, 4), "k_tw_q": (5, 6), "k_tw_va": (3, 3),
    "k_tw_vr": (4, 4), "k_stress_diag": (7, 7), "k_stress_rphi": (5, 5), "k_visc_va": (4, 4),
    "k_visc_vr": (5, 5), "k_cfl_cells": (4, 7), "k_pressure": (3, 2), "k_potential": (2, 2),
    "k_ring_mean": (1, 1),
}
STEP_BYTES = (256, 320)
# share of SURVEY.md 8(d)'s pass model (doubles per cell, isothermal | adiabatic) that a fused kernel stands for
MODEL_PASSES = {
    "k_transport_fused": ("B+C+D", (27, 33)),
    "k_transport_theta_march": ("C+D", (19, 23)),
    "k_source_march": ("A", (5, 7)),
}


def affinity_threads(cap=16):
    return max(1, min(cap, len(os.sched_getaffinity(0))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nr", type=int, default=NR_PER_GPU, help="rings per GPU")
    ap.add_argument("--nphi", type=int, default=NPHI)
    ap.add_argument("--eos", choices=["isothermal", "ideal"], default="isothermal",
                    help="ideal: BASELINE config 3 physics (energy equation, viscous heating) on the same grid")
    args = ap.parse_args()

    # the contract is ONE JSON line on stdout: libraries that print there (RCCL's version banner at
    # communicator creation) are sent to stderr for the whole run, the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("RCCL_LOG_LEVEL", "0")

    os.environ.setdefault("OMP_NUM_THREADS", str(affinity_threads()))
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

    import numpy as np
    import torch  # first: the HIP runtime torch bundles must be the one the library binds to
    import torch.distributed as dist

    import fargocpt_amd
    from fargocpt_amd import binding as B, driver, setups

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("FCPT_BENCH_FORCE_DIST") == "1"  # 1-rank RCCL group: rehearsal
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)
        # the step's kernels on a side stream: on the null stream they shared a hardware queue with RCCL's
        # stream, and the CFL kernels queued behind the ghost exchange (fcpt_cfl_begin) could not overlap it
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))

    lib = fargocpt_amd.load()
    nr_global = args.nr * world
    d = setups.planet_disk(lib, nr_global, args.nphi, adiabatic=args.eos == "ideal")
    if world > 1:
        # weak scaling: keep dr/r of the 1-GPU grid, extend the disk outward
        d.rmax = d.rmin * (2.5 / 0.4) ** world
        d.damping_time_radius_outer = d.rmax
    d.rank, d.nranks = rank, world
    bodies = setups.jupiter_bodies(d)

    radii = lib.radii(d)
    fields = lib.initial_fields(d.copy(), radii)  # slab-local
    ctx = driver.make_context(lib, d, fields=fields, radii=radii, bodies=bodies)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    adi = 1 if d.eos == B.EOS_IDEAL else 0

    # ---- one step ------------------------------------------------------------
    if not use_dist:
        def run(n):
            ctx.run_steps(n, snap=False)  # dt stays on the device, no host sync inside

        def pre_loop():
            for _ in range(2):
                ctx.calculate_timestep(ctx.cfl())
    else:
        from fargocpt_amd.parallel import DistributedSlab
        slab = DistributedSlab(ctx, device=dev)

        def run(n):
            for _ in range(n):
                slab.step_async()  # CFL -> all_reduce(MIN) -> policy -> step -> 7-ring exchange -> post

        def pre_loop():
            slab.prepare()

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    # sim::init's pre-loop time-step calls (main.cpp:117, simulation.cpp:466-468)
    pre_loop()

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
    ctx.profile_start([names.index(dominant)], max_launches=args.steps + 8)
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    if use_dist:
        dist.barrier()
    t1 = time.perf_counter()
    dom = ctx.profile_stop()[dominant]
    elapsed = t1 - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = ctx.state()
    finite = all(np.isfinite(v).all() for v in st.values())

    if rank == 0:
        cells = nr_global * args.nphi
        value = cells * args.steps / elapsed
        dom_ms = dom[0] / max(1, dom[1])
        slab_cells = ctx.nr * args.nphi
        algo_bytes = ALGO_DOUBLES.get(dominant, (0, 0))[adi] * 8 * slab_cells
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = valu_busy = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("workload") == f"{args.nr}x{args.nphi}":
                    traffic = rec.get("hbm_bytes_per_launch", {}).get(dominant)
                    valu_busy = rec.get("valu_busy", {}).get(dominant)
            except Exception:
                traffic = valu_busy = None
        out = {
            "metric": "cell-updates/s on Nr x Nphi polar grid", "value": value, "unit": "cell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{nr_global}x{args.nphi} {'ideal-gas' if adi else 'locally-isothermal'} disk + 1 Jupiter-mass planet "
                                   "(examples/config.yml physics: alpha=1e-3, TW artificial viscosity, "
                                   "reflecting BC + damping, FARGO transport, Euler), "
                                   f"{args.nr} rings per GPU",
                       "grid": [nr_global, args.nphi], "parallelism": f"radial slabs x{world}",
                       "finite": bool(finite)},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": dom_ms, "launches": dom[1],
                         "algorithmic_bytes_per_launch": algo_bytes,
                         # the whole step against SURVEY.md 8(d)'s 256|320 B per cell-update
                         "step_frac": value * STEP_BYTES[adi] / (HBM_PEAK_GBS * 1e9 * world),
                         # busy fraction of the FP64 vector ALUs in this kernel (PMC, profiles/pmc_latest.json):
                         # k_transport_fused moves 48 B per cell instead of the model's 216 B and is bound by
                         # the vector pipeline, not by HBM
                         "valu_busy": valu_busy,
                         # the same kernel time priced with the contract's pass model instead of the kernel's own
                         # minimal traffic: what fraction of the HBM peak the unfused passes it replaces would need
                         "model": ({"passes": MODEL_PASSES[dominant][0],
                                    "bytes_per_launch": MODEL_PASSES[dominant][1][adi] * 8 * slab_cells,
                                    "achieved": MODEL_PASSES[dominant][1][adi] * 8 * slab_cells / (dom_ms * 1e-3) / 1e9,
                                    "frac": MODEL_PASSES[dominant][1][adi] * 8 * slab_cells / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                   if dominant in MODEL_PASSES and dom_ms > 0 else None),
                         "note": ("k_transport_fused does passes B+C+D of SURVEY.md 8(d)'s model (27|33 doubles per cell) "
                                  "with 6|8 doubles of traffic; it is bound by the FP64 vector ALUs (valu_busy), not by HBM: "
                                  "frac is its HBM fraction, step_frac the whole step against the 256|320 B model")
                         if dominant == "k_transport_fused" else None},
            "kernel_ms_per_step": {k: v[0] / cal for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lib, d, fields, radii, bodies)
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    ctx.close()
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(lib, d, fields, radii, bodies):
    """The CPU oracle (oracle/fargo_oracle.c, a C+OpenMP restatement of the reference loops)
    timed on this box's host cores on a bounded sample of the same workload."""
    import ctypes
    import subprocess
    from fargocpt_amd import binding as B, driver

    so = os.path.join(ROOT, "oracle", "libfargo_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    orc = B.Library(ctypes.CDLL(so), "orc_")
    threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    ctx = driver.make_context(orc, d, fields=fields, radii=radii, bodies=bodies)
    for _ in range(2):
        ctx.calculate_timestep(ctx.cfl())
    ctx.run_steps(1)  # page in
    t0 = time.perf_counter()
    n = 0
    while True:
        ctx.run_steps(1)
        n += 1
        el = time.perf_counter() - t0
        if el > 12.0 or n >= 40:
            break
    cells = d.nr_global * d.nphi
    ctx.close()
    return {"value": cells * n / el, "unit": "cell-updates/s", "cores": threads, "kind": "port",
            "sample": f"{n} steps of the same {d.nr_global}x{d.nphi} workload, oracle/fargo_oracle.c "
                      f"(-O2, OpenMP, {threads} threads)"}


if __name__ == "__main__":
    main()
