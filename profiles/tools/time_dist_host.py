"""Host cost of the N>1 step loop, rehearsed on ONE GPU: a 1-rank RCCL group, the slab built as the middle one
of three (both neighbours present), ghost rings sent to itself.  On a small grid the wall time per step is the
host-side enqueue cost (ctypes + torch.distributed), on the bench grid it shows what the loop adds to the GPU time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
import fargocpt_amd
from fargocpt_amd import binding as B, driver, setups
from fargocpt_amd.parallel import DistributedSlab

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
lib = fargocpt_amd.load()
# FCPT_SIDE_STREAM=n: run the loop on a torch side stream (the n-th created) instead of the null stream
_n = int(os.environ.get("FCPT_SIDE_STREAM", "0"))
if _n > 0:
    _streams = [torch.cuda.Stream(device=dev) for _ in range(_n)]
    torch.cuda.set_stream(_streams[-1])


class SelfSlab(DistributedSlab):
    def __init__(self, ctx):
        super().__init__(ctx, device=dev)
        cnt = ctx.exchange_count()
        mk = lambda: torch.zeros(cnt, dtype=torch.float64, device=dev)
        self.has_inner = self.has_outer = True
        self.s_in, self.r_in, self.s_out, self.r_out = mk(), mk(), mk(), mk()
        self.world = 2   # take the communicating branches
        self.rank = 0

    def exchange(self, overlap=None):
        self.ctx.exchange_pack(self._arg(self.s_in), self._arg(self.s_out))
        ops = [dist.P2POp(dist.isend, self.s_in, 0), dist.P2POp(dist.irecv, self.r_in, 0),
               dist.P2POp(dist.isend, self.s_out, 0), dist.P2POp(dist.irecv, self.r_out, 0)]
        works = dist.batch_isend_irecv(ops)
        if overlap is not None:
            overlap()
        for w in works:
            w.wait()
        self.r_in.copy_(self.s_in)   # keep the slab's own ghosts: physics stays sane
        self.r_out.copy_(self.s_out)
        self.ctx.exchange_unpack(self._arg(self.r_in), self._arg(self.r_out))


GRIDS = ((2048, 4096, 30),) if os.environ.get("FCPT_TRACE") else ((64, 256, 200), (2048, 4096, 100))
for nr, nphi, steps in GRIDS:
    d = setups.planet_disk(lib, 3 * nr, nphi)
    d.rank, d.nranks = 1, 3
    radii = lib.radii(d)
    fields = lib.initial_fields(d.copy(), radii)
    ctx = driver.make_context(lib, d, fields=fields, radii=radii, bodies=setups.jupiter_bodies(d))
    slab = SelfSlab(ctx)
    for _ in range(2):
        slab.ctx.calculate_timestep(slab.ctx.cfl())
    for _ in range(10):
        slab.step_async()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        slab.step_async()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # the same loop without communication
    t1 = time.perf_counter()
    for _ in range(steps):
        ctx.cfl_device(slab._dt.data_ptr())
        ctx.calculate_timestep_device(slab._dt.data_ptr())
        ctx.step_device()
        ctx.post_device()
    torch.cuda.synchronize()
    t_nocomm = time.perf_counter() - t1
    print(f"{nr}x{nphi}: host enqueue {1e6 * t_host / steps:.0f} us/step, wall {1e6 * t_all / steps:.0f} us/step, "
          f"no-communication loop {1e6 * t_nocomm / steps:.0f} us/step", flush=True)
    ctx.close()
dist.destroy_process_group()
