"""The N > 1 path on CPU: one process per radial slab, torch.distributed with the gloo
backend, world_size 2.  Slab contexts are oracle contexts (there is no GPU here); the
orchestration under test -- radial split, dt all-reduce(MIN), 7-ring neighbour exchange with
isend/irecv, global gather -- is the code bench.py runs with the nccl backend."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nsteps, out):
    os.environ["OMP_NUM_THREADS"] = "2"
    sys.path.insert(0, ROOT)
    import ctypes
    import fargocpt_amd
    from fargocpt_amd import binding as B, driver, setups
    from fargocpt_amd.parallel import DistributedSlab

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lib = fargocpt_amd.load()
    orc = B.Library(ctypes.CDLL(os.path.join(ROOT, "oracle", "libfargo_oracle.so")), "orc_")
    d = setups.planet_disk(lib, 64, 48, adiabatic=True)
    dfull = d.copy()
    radii = lib.radii(dfull)
    fields = lib.initial_fields(dfull, radii)
    d.sigma0 = dfull.sigma0
    d.rank, d.nranks = rank, world
    s = lib.split_domain(d)
    sub = tuple(np.ascontiguousarray(f[s.imin:s.imin + s.nr + (1 if k == 1 else 0)]) for k, f in enumerate(fields))
    ctx = driver.make_context(orc, d, fields=sub, radii=radii, bodies=setups.jupiter_bodies(d))
    slab = DistributedSlab(ctx)
    slab.prepare()
    dts = [slab.step() for _ in range(nsteps)]
    glob = slab.gather()  # on rank 0
    if rank == 0:
        np.savez(out, dts=np.array(dts), **glob)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_gloo_match_single_slab(tmp_path, product, oracle):
    from fargocpt_amd import driver, setups
    from tests.util import rel_err
    nsteps = 15
    out = str(tmp_path / "dist.npz")
    mp.start_processes(_worker, args=(2, _free_port(), nsteps, out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    # single-slab oracle run of the same problem
    d = setups.planet_disk(product, 64, 48, adiabatic=True)
    ctx = driver.make_context(oracle, d, bodies=setups.jupiter_bodies(d))
    s = driver.SlabSet([ctx])
    s.prepare()
    dts = s.run(nsteps)
    ref = s.gather()
    assert np.array_equal(got["dts"], np.array(dts))
    for k in ("sigma", "vrad", "vazi", "energy"):
        assert got[k].shape == ref[k].shape
        assert rel_err(got[k], ref[k]) == 0.0, k
