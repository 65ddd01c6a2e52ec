"""The N > 1 path with HIP slabs: two processes, one radial slab each, both on the one GPU of the
test box, torch.distributed over gloo (RCCL refuses two ranks on one device; the ghost rings are
staged through the host, everything else -- split, device-resident dt with an all-reduce(MIN) of a
device scalar, pack / unpack kernels, post -- is the code path bench.py runs over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nsteps, adiabatic, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import fargocpt_amd
    from fargocpt_amd import driver, setups
    from fargocpt_amd.parallel import DistributedSlab

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lib = fargocpt_amd.load()
    d = setups.planet_disk(lib, 72, 320, adiabatic=adiabatic)
    dfull = d.copy()
    radii = lib.radii(dfull)
    fields = lib.initial_fields(dfull, radii)
    d.sigma0 = dfull.sigma0
    d.rank, d.nranks = rank, world
    s = lib.split_domain(d)
    sub = tuple(np.ascontiguousarray(f[s.imin:s.imin + s.nr + (1 if k == 1 else 0)]) for k, f in enumerate(fields))
    ctx = driver.make_context(lib, d, fields=sub, radii=radii, bodies=setups.jupiter_bodies(d))
    slab = DistributedSlab(ctx, device=torch.device("cuda", 0))
    slab.prepare()
    for _ in range(nsteps):
        slab.step_async()
    ctx.synchronize()
    glob = slab.gather()  # on rank 0
    if rank == 0:
        np.savez(out, time=ctx.clock.time, **glob)
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("adiabatic", [False, True])
def test_two_hip_slabs_over_torch_distributed(tmp_path, product, adiabatic):
    from fargocpt_amd import driver, setups
    from tests.util import rel_err
    nsteps = 12
    out = str(tmp_path / "dist.npz")
    mp.start_processes(_worker, args=(2, _free_port(), nsteps, adiabatic, out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    d = setups.planet_disk(product, 72, 320, adiabatic=adiabatic)
    ctx = driver.make_context(product, d, bodies=setups.jupiter_bodies(d))
    s = driver.SlabSet([ctx])
    s.prepare()
    s.run(nsteps)
    ref = s.gather()
    assert abs(float(got["time"]) - ctx.clock.time) <= 1e-12 * ctx.clock.time
    for k in ("sigma", "vrad", "vazi") + (("energy",) if adiabatic else ()):
        assert got[k].shape == ref[k].shape
        # the reference itself agrees to 4e-13 between 1 and 2 ranks
        assert rel_err(got[k], ref[k]) <= 1e-12, k
    ctx.close()


# ---- RCCL inside the library (fcpt_comm_init / fcpt_exchange / fcpt_cfl_allreduce / fcpt_run_steps) ----------
# One GPU cannot hold two RCCL ranks, so the transfer itself is rehearsed with a communicator of one rank whose
# slab is its own neighbour on both sides (option comm_loopback): rows [7,14) arrive in rows [0,7) and rows
# [nr-14,nr-7) in rows [nr-7,nr), through ncclSend/ncclRecv on the context's stream.

def _middle_slab(product, adiabatic, loopback):
    import torch  # the HIP runtime torch bundles first
    from fargocpt_amd import driver, setups
    d = setups.planet_disk(product, 3 * 40, 320, adiabatic=adiabatic)
    d.rank, d.nranks = 1, 3
    radii = product.radii(d)
    fields = product.initial_fields(d.copy(), radii)
    from tests.util import perturb
    fields = perturb(fields, d, 1e-3)
    ctx = driver.make_context(product, d, fields=fields, radii=radii, bodies=setups.jupiter_bodies(d))
    if loopback:
        ctx.set_option("comm_loopback", 1)
        ctx.comm_init(product.comm_unique_id())
    return ctx


def _host_loopback_exchange(ctx):
    cnt = ctx.exchange_count()
    s_in, s_out = np.zeros(cnt), np.zeros(cnt)
    ctx.exchange_pack(s_in, s_out)
    ctx.synchronize()
    ctx.exchange_unpack(s_in, s_out)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("adiabatic,overlap", [(False, 0), (False, 1), (True, 0), (True, 1)])
def test_library_rccl_exchange_and_loop(product, adiabatic, overlap):
    """fcpt_exchange + fcpt_cfl_allreduce + the multi-slab fcpt_run_steps against the same sequence driven from the
    host with host-staged ghost rings: identical bits (state, clock), with and without the transfers overlapping
    the interior CFL."""
    nsteps = 8
    a = _middle_slab(product, adiabatic, loopback=True)
    a.set_option("comm_overlap", overlap)
    assert a.get_option("comm_overlap") == overlap
    b = _middle_slab(product, adiabatic, loopback=False)
    # sim::init's sequence
    for ctx, xchg in ((a, a.exchange), (b, lambda: _host_loopback_exchange(b))):
        ctx.calculate_timestep(ctx.cfl_allreduce() if ctx is a else ctx.cfl())
        xchg()
        ctx.apply_boundary(0.0, False)
        ctx.calculate_timestep(ctx.cfl_allreduce() if ctx is a else ctx.cfl())
        xchg()
    assert a.run_steps(nsteps) == nsteps
    for _ in range(nsteps):
        dt = b.calculate_timestep(b.cfl())
        b.step(dt)
        _host_loopback_exchange(b)
        b.post(dt)
    ca, cb = a.clock, b.clock
    assert (ca.time, ca.last_dt, ca.n_hydro_iter) == (cb.time, cb.last_dt, cb.n_hydro_iter)
    sa, sb = a.state(), b.state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    a.comm_destroy()
    a.close()
    b.close()


def test_library_comm_errors(product):
    from fargocpt_amd import binding as B
    ctx = _middle_slab(product, False, loopback=False)
    with pytest.raises(B.FcptError, match="fcpt_comm_init"):
        ctx.exchange()
    with pytest.raises(B.FcptError, match="unknown option"):
        ctx.set_option("no_such_switch", 1)
    ctx.close()


def test_bench_n_rank_line_rehearsed_on_one_gpu(tmp_path):
    """`bench.py --gpus 2 --transport host`: every code path of the N-rank bench line -- the launcher, the slab
    contexts, the communicator hand-shake, the settle protocol with its all-reduced stop decision, barriers and
    max-over-ranks timing, the strong-scaling row (BASELINE config 4) and the cross-rank check of ghost rows and clocks --
    with the ranks sharing this box's GPU: gloo process group + the library's host-staged transport instead of RCCL
    (which refuses two ranks on one device).  The line says that it is a rehearsal, not a scaling number."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["FCPT_BENCH_SETTLE_STEPS"], env["FCPT_BENCH_SETTLE_MS"] = "10", "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--transport", "host", "--steps", "6",
                        "--warmup", "2", "--nr", "64", "--nphi", "512", "--settle-blocks", "1"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["grid"] == [128, 512]
    assert rec["config"]["ghost_rings_and_clock_consistent"] is True and rec["config"]["finite"] is True
    assert "NOT a scaling number" in rec["config"]["rehearsal"] and "host-staged" in rec["config"]["communication"]
    assert len(rec["ms_per_step_per_rank"]) == 2 and rec["roofline"]["frac"] <= 1.0
    s = rec["strong_scaling"]
    assert s["scaling"] == "strong" and s["grid"] == [2048, 6144] and s["ghost_rings_and_clock_consistent"] is True
    assert s["value"] == pytest.approx(2048 * 6144 / (s["ms_per_step"] * 1e-3), rel=1e-9)
