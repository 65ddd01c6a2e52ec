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
        grids = [("sigma", B.F_SIGMA), ("vrad", B.F_VRAD), ("vazi", B.F_VAZI), ("energy", B.F_ENERGY)]
        if self.ctxs[0].desc.write_massflow:
            grids.append(("massflow", B.F_MASSFLOW))
        for name, f in grids:
            parts = []
            for c in self.ctxs:
                a = c.download(f)
                s = c.split
                lo = 0 if s.is_first else B.OVERLAP
                hi = a.shape[0] - (0 if s.is_last else B.OVERLAP)
                if f in B.VECTOR_FIELDS and not s.is_last:
                    hi -= 1
                parts.append(a[lo:hi])
            out[name] = np.concatenate(parts, axis=0)
        return out


class CircularOrbits:
    """The N-body stand-in of the host driver (csrc/host/fargocpt_hip_main.cpp, set_bodies): the star at the origin,
    every planet on its initial circular orbit seen in the frame rotating with OmegaFrame, its mass ramped up as
    t_planet::get_rampup_mass does (nbody/planet.cpp:166-179), and the indirect term of the star-centred frame
    without the disk (refframe::IndirectTermPlanets, frame_of_reference.cpp:138-160: minus the star's velocity
    change over the step divided by dt; for circular orbits the time average of G m r_p / a^3 over [t, t + dt])."""

    def __init__(self, d: B.Desc, bodies):
        self.d = d
        self.bodies = list(bodies)   # (semi-major axis, mass, ramp-up time in orbital periods)

    def at(self, t: float, dt: float):
        import math
        d = self.d
        x, y, m = [], [], []
        itx = ity = 0.0
        for k, (a, mass, ramp) in enumerate(self.bodies):
            om = math.sqrt(d.G * (d.hydro_center_mass + mass) / (a * a * a)) if a > 0 else 0.0
            ang = (om - d.omega_frame) * t
            x.append(a * math.cos(ang))
            y.append(a * math.sin(ang))
            mk = mass
            if ramp > 0 and om > 0:
                period = 2 * math.pi / om
                if t < ramp * period:
                    cs = math.cos(t * (math.pi / 2) / (ramp * period))
                    mk = mass * (1.0 - cs * cs)
            m.append(mk)
            if k > 0 and a > 0 and dt > 0:
                g, w = d.G * mass / (a * a), om * dt
                itx -= g * (math.sin(ang + w) - math.sin(ang)) / w
                ity -= g * (math.cos(ang) - math.cos(ang + w)) / w
        return x, y, m, None, (itx, ity)


def run_to_snapshots(ctx: B.Context, d: B.Desc, orbits: "CircularOrbits | None" = None, on_snapshot=None,
                     max_snapshots=None) -> int:
    """sim::run (simulation.cpp:505-558) for one slab with moving bodies: CFL, CalculateTimeStep, monitor-time
    snapping, bodies at the step's start, step, post; `on_snapshot(n)` at every Nmonitor-th monitor time.
    Returns the number of hydro steps."""
    s = SlabSet([ctx])
    s.prepare()
    steps, n_monitor = 0, 0
    nsnap = d.nsnapshots if max_snapshots is None else min(d.nsnapshots, max_snapshots)
    t_final = nsnap * d.nmonitor * d.monitor_timestep
    time = 0.0
    while time < t_final:
        cfl_dt = ctx.calculate_timestep(ctx.cfl())
        step_dt = ctx.snap_to_monitor(cfl_dt)
        t_next = (n_monitor + 1) * d.monitor_timestep
        if orbits is not None and len(orbits.bodies) > 1:
            ctx.set_bodies(*orbits.at(time, step_dt))
        ctx.step(step_dt)
        ctx.post(step_dt)
        time += step_dt
        steps += 1
        if abs(t_next - time) < 1e-6 * cfl_dt:
            n_monitor += 1
            c = ctx.clock
            c.n_monitor = n_monitor
            c.n_snapshot = n_monitor // d.nmonitor
            ctx.clock = c
            if n_monitor % d.nmonitor == 0 and on_snapshot is not None:
                on_snapshot(n_monitor // d.nmonitor)
    return steps
