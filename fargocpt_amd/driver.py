"""Run-loop plumbing over the C ABI: the call sequence of the reference's
main()/sim::run (src/main.cpp:117-158, src/simulation.cpp:462-553) for one
slab or for a set of radial slabs.  Works with any Library exporting the ABI.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np

from . import binding as B


def make_context(lib: B.Library, d: B.Desc, fields=None, radii=None, bodies=None, irradiation=None) -> B.Context:
    """Create a slab context, upload initial fields (generated when not given)
    and run init_physics."""
    d = d.copy()
    if radii is None:
        radii = lib.radii(d)
    if fields is None:
        fields = lib.initial_fields(d, radii)  # may rescale d.sigma0 (SetSigma0)
    ctx = lib.create(d, radii)
    sigma, vrad, vazi, energy = fields
    ctx.upload(B.F_SIGMA, sigma)
    ctx.upload(B.F_VRAD, vrad)
    ctx.upload(B.F_VAZI, vazi)
    ctx.upload(B.F_ENERGY, energy)
    if bodies is not None:
        ctx.set_bodies(*bodies)
    if irradiation is not None:  # (temperature[], radius[], rampup_time[] | None) per body
        ctx.set_body_irradiation(*irradiation)
    ctx.init_physics()
    return ctx


class SlabSet:
    """One or more radial slabs advanced in lock step.

    `allreduce_min(x) -> float` and `exchange(slab_index, send_inner, send_outer)
    -> (recv_inner, recv_outer)` abstract the two communication points of the
    path (cfl.cpp:379 and commbound.cpp:130-158); the defaults serve slabs
    that all live in this process.
    """

    def __init__(self, ctxs: Sequence[B.Context],
                 allreduce_min: Optional[Callable[[float], float]] = None):
        self.ctxs: List[B.Context] = list(ctxs)
        self._allreduce_min = allreduce_min or (lambda x: x)
        self.dt_scale = 1.0  # tests: step with a multiple of the CFL time step

    # -- communication --------------------------------------------------------
    def global_cfl(self) -> float:
        local = min(c.cfl() for c in self.ctxs)
        return self._allreduce_min(local)

    def exchange_local(self):
        """CommunicateBoundaries between slabs held in this process (host staging)."""
        n = len(self.ctxs)
        if n < 2:
            return
        cnt = self.ctxs[0].exchange_count()
        bufs = [(np.zeros(cnt), np.zeros(cnt)) for _ in range(n)]
        for k, c in enumerate(self.ctxs):
            c.exchange_pack(bufs[k][0] if k > 0 else None, bufs[k][1] if k < n - 1 else None)
        for k, c in enumerate(self.ctxs):
            c.exchange_unpack(bufs[k - 1][1] if k > 0 else None,
                              bufs[k + 1][0] if k < n - 1 else None)

    # -- reference call sequence ---------------------------------------------
    def calculate_timestep(self) -> float:
        g = self.global_cfl()
        dts = [c.calculate_timestep(g) for c in self.ctxs]
        return dts[0]

    def prepare(self):
        """main(): CalculateTimeStep, CommunicateBoundariesAll (main.cpp:117,147);
        sim::init(): BC(final=false), CalculateTimeStep, CommunicateBoundaries
        (simulation.cpp:462-474)."""
        self.calculate_timestep()
        self.exchange_local()
        for c in self.ctxs:
            c.apply_boundary(0.0, False)
        self.calculate_timestep()
        self.exchange_local()

    def step(self, snap: bool = False) -> float:
        dt = self.calculate_timestep()
        step_dt = self.ctxs[0].snap_to_monitor(dt) if snap else dt
        step_dt *= self.dt_scale
        for c in self.ctxs:
            c.step(step_dt)
        self.exchange_local()
        for c in self.ctxs:
            c.post(step_dt)
        return step_dt

    def run(self, nsteps: int, snap: bool = False):
        dts = []
        for _ in range(nsteps):
            dts.append(self.step(snap))
        return dts

    def gather(self):
        """Global fields with overlap rings stripped (write2D, polargrid.cpp:135-180)."""
        out = {}
        for name, f in (("sigma", B.F_SIGMA), ("vrad", B.F_VRAD), ("vazi", B.F_VAZI),
                        ("energy", B.F_ENERGY)):
            parts = []
            for c in self.ctxs:
                a = c.download(f)
                s = c.split
                lo = 0 if s.is_first else B.OVERLAP
                hi = a.shape[0] - (0 if s.is_last else B.OVERLAP)
                if f == B.F_VRAD and not s.is_last:
                    hi -= 1
                parts.append(a[lo:hi])
            out[name] = np.concatenate(parts, axis=0)
        return out
