"""One radial slab per process over torch.distributed (RCCL on MI355X: backend "nccl";
CPU tests: "gloo").  The two communication points of the path are the reference's:

* cfl.cpp:379          MPI_Allreduce(MIN) of the CFL time step  -> dist.all_reduce(MIN)
* commbound.cpp:98-182 7 overlap rings of Sigma, v_r, v_phi(, e) with both radial neighbours
                        -> dist.batch_isend_irecv of one packed buffer per direction

Buffers live where the slab's library keeps its grids: torch CUDA tensors for the HIP
library (their device addresses are handed to fcpt_exchange_pack/unpack, kernels and RCCL
share torch's current stream), CPU tensors for host libraries.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from . import binding as B


class DistributedSlab:
    def __init__(self, ctx: B.Context, device: torch.device | None = None):
        self.ctx = ctx
        ready = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if ready else 0
        self.world = dist.get_world_size() if ready else 1
        self.device = device or torch.device("cpu")
        self.on_gpu = self.device.type == "cuda"
        cnt = ctx.exchange_count()
        mk = lambda: torch.zeros(cnt, dtype=torch.float64, device=self.device)
        self.has_inner, self.has_outer = self.rank > 0, self.rank < self.world - 1
        self.s_in, self.r_in = (mk(), mk()) if self.has_inner else (None, None)
        self.s_out, self.r_out = (mk(), mk()) if self.has_outer else (None, None)
        self._dt = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._cfl_ready = False  # self._dt holds the reduced CFL step of the current state (step_async)
        # gloo has no send/recv of device tensors: GPU slabs on a gloo group (tests: several ranks on
        # the one GPU of a box) stage the ghost rings through host tensors; RCCL sends them in place
        self.stage_host = self.on_gpu and ready and dist.get_backend() == "gloo"
        self.split_step = False
        if self.on_gpu:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            ctx.set_stream(stream)
            # fcpt_step_device_begin/_end (the whole exchange under the interior chunks of the transport) is opt-in:
            # in the one-GPU rehearsal (profiles/tools/time_dist_host.py) it measured the same 470 us per step as
            # the plain step with the CFL overlap -- the cost left is stream-synchronisation latency, not transfer
            # time -- and 543 us on the null stream, where the library's side stream and RCCL's serialise
            self.split_step = stream != 0 and os.environ.get("FCPT_SPLIT_STEP") == "1"

    def _arg(self, t):
        if t is None:
            return None
        return t.data_ptr() if self.on_gpu else t.numpy()

    # cfl.cpp:379
    def global_cfl(self) -> float:
        self._cfl_ready = False
        self._dt[0] = self.ctx.cfl()
        if self.world > 1:
            dist.all_reduce(self._dt, op=dist.ReduceOp.MIN)
        return float(self._dt.item())

    def calculate_timestep(self) -> float:
        return self.ctx.calculate_timestep(self.global_cfl())

    # commbound.cpp:98-182
    def exchange(self, overlap=None):
        """`overlap`: work queued on the compute stream between posting the transfers and waiting for them
        (RCCL runs them on its own stream): it must not touch the ghost rings."""
        if self.world == 1:
            return
        self.ctx.exchange_pack(self._arg(self.s_in), self._arg(self.s_out))
        if self.stage_host:
            self.ctx.synchronize()
            s_in, s_out = (t.cpu() if t is not None else None for t in (self.s_in, self.s_out))
            r_in, r_out = (torch.empty_like(t) if t is not None else None for t in (s_in, s_out))
        else:
            s_in, s_out, r_in, r_out = self.s_in, self.s_out, self.r_in, self.r_out
        ops = []
        if self.has_inner:
            ops += [dist.P2POp(dist.isend, s_in, self.rank - 1), dist.P2POp(dist.irecv, r_in, self.rank - 1)]
        if self.has_outer:
            ops += [dist.P2POp(dist.isend, s_out, self.rank + 1), dist.P2POp(dist.irecv, r_out, self.rank + 1)]
        works = dist.batch_isend_irecv(ops)
        if overlap is not None:
            overlap()
        for w in works:
            w.wait()
        if self.stage_host:
            if self.has_inner:
                self.r_in.copy_(r_in)
            if self.has_outer:
                self.r_out.copy_(r_out)
        self.ctx.exchange_unpack(self._arg(self.r_in), self._arg(self.r_out))

    def step_async(self):
        """One step with dt kept on the device (GPU slabs only): policy kernel, step, exchange, post, then the
        CFL kernels and the MIN all-reduce (cfl.cpp:379) of a one-element device tensor FOR THE NEXT STEP -- all
        enqueued on the current stream, no host synchronisation.  The ghost rings travel while the interior
        chunks of the transport and the CFL reduction over the interior rings run (fcpt_step_device_begin/_end,
        fcpt_cfl_begin).  `invalidate()` after touching the fields
        by other means."""
        assert self.on_gpu
        if not self._cfl_ready:
            self._reduce_cfl()
        self.ctx.calculate_timestep_device(self._dt.data_ptr())
        if self.world > 1 and self.split_step:
            # the chunks holding the neighbours' ghost rings first; the others run on the library's side stream
            # under the pack kernel and the transfers
            self.ctx.step_device_begin()
            self.exchange(overlap=self._finish_step)
        elif self.world > 1:
            self.ctx.step_device()
            self.exchange(overlap=self.ctx.cfl_begin)
        else:
            self.ctx.step_device()
        self.ctx.post_device()
        self._reduce_cfl()

    def _finish_step(self):
        self.ctx.step_device_end()
        self.ctx.cfl_begin()

    def _reduce_cfl(self):
        self.ctx.cfl_device(self._dt.data_ptr())
        if self.world > 1:
            dist.all_reduce(self._dt, op=dist.ReduceOp.MIN)
        self._cfl_ready = True

    def invalidate(self):
        """The reduced CFL step held for the next step_async is stale (fields changed outside step_async)."""
        self._cfl_ready = False

    def prepare(self):
        """main.cpp:117,147 and sim::init (simulation.cpp:462-474)."""
        self.calculate_timestep()
        self.exchange()
        self.ctx.apply_boundary(0.0, False)
        self.calculate_timestep()
        self.exchange()

    def step(self, snap: bool = False) -> float:
        dt = self.calculate_timestep()
        step_dt = self.ctx.snap_to_monitor(dt) if snap else dt
        self.ctx.step(step_dt)
        self.exchange()
        self.ctx.post(step_dt)
        return step_dt

    def gather(self):
        """Global grids on rank 0 with the overlap rings stripped (write2D windows,
        polargrid.cpp:135-180); other ranks return None."""
        s = self.ctx.split
        out = {}
        for name, f in (("sigma", B.F_SIGMA), ("vrad", B.F_VRAD), ("vazi", B.F_VAZI), ("energy", B.F_ENERGY)):
            a = self.ctx.download(f)
            lo = 0 if s.is_first else B.OVERLAP
            hi = a.shape[0] - (0 if s.is_last else B.OVERLAP)
            if f == B.F_VRAD and not s.is_last:
                hi -= 1
            part = torch.from_numpy(np.ascontiguousarray(a[lo:hi]))
            if self.world == 1:
                out[name] = part.numpy()
                continue
            sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
            dist.all_gather(sizes, torch.tensor([part.shape[0]], dtype=torch.int64))
            nmax = max(int(n.item()) for n in sizes)
            padded = torch.zeros(nmax, part.shape[1], dtype=torch.float64)
            padded[:part.shape[0]] = part
            parts = [torch.zeros_like(padded) for _ in sizes]
            dist.all_gather(parts, padded)  # equal-sized pieces, trimmed below
            out[name] = torch.cat([p[:int(n.item())] for p, n in zip(parts, sizes)], 0).numpy()
        return out if self.rank == 0 else None
