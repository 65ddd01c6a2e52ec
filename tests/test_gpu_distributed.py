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
