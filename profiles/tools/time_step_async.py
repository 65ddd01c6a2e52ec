#!/usr/bin/env python3
"""Host-side cost of one step: time to ENQUEUE steps vs time until the GPU has finished them, for the
C loop (fcpt_run_steps) and for DistributedSlab.step_async on a 1-rank RCCL group."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
import fargocpt_amd
from fargocpt_amd import driver, setups
from fargocpt_amd.parallel import DistributedSlab

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
lib = fargocpt_amd.load()
d = setups.planet_disk(lib, 2048, 4096)
ctx = driver.make_context(lib, d, bodies=setups.jupiter_bodies(d))
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
slab = DistributedSlab(ctx, device=dev)
slab.prepare()
def timed(name, fn, n=200):
    fn(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: enqueue {1e3*(t1-t0)/n:.3f} ms/step, complete {1e3*(t2-t0)/n:.3f} ms/step")
timed("fcpt_run_steps (C loop)", lambda n: ctx.run_steps(n, snap=False))
timed("step_async (python)", lambda n: [slab.step_async() for _ in range(n)])
def pieces(n):
    for _ in range(n):
        ctx.cfl_device(slab._dt.data_ptr())
timed("cfl_device only", pieces)
ctx.close(); dist.destroy_process_group()
