"""Seeded random walks through the configuration space of the path: every draw is one descriptor (grid, EOS,
viscosity, artificial viscosity, integrator, limiter, transport, per-variable boundary conditions, damping
targets, cooling, StabilizeViscosity, slabs, planet) advanced a few steps by the HIP library and by the oracle.
The hand-written parity tests pin named combinations; this one looks for the combinations nobody named."""
import os

import numpy as np
import pytest

from fargocpt_amd import binding as B, setups
from tests.util import rel_err, run_pair

pytestmark = pytest.mark.gpu
TOL = 1e-10
NSEEDS = int(os.environ.get("FCPT_FUZZ_SEEDS", "160"))
WIDE = os.environ.get("FCPT_FUZZ_WIDE") == "1"   # exploratory runs: arbitrary ring lengths and ring counts
FIRST = int(os.environ.get("FCPT_FUZZ_FIRST", "0"))


_EXTRA = {"dt_scale": 1.0}


def draw(lib, seed):
    rng = np.random.default_rng(1000 + seed)
    pick = lambda *xs: xs[int(rng.integers(len(xs)))]
    adiabatic = bool(rng.integers(2))
    nr, nphi = pick(16, 23, 40, 57, 72), pick(8, 17, 64, 130, 144, 256, 320, 514)
    if WIDE:
        nr, nphi = int(rng.integers(16, 121)), int(rng.integers(2, 1500))
    d = setups.planet_disk(lib, nr, nphi, adiabatic=adiabatic)
    d.radial_spacing = pick(B.SPACING_LOGARITHMIC, B.SPACING_LOGARITHMIC, B.SPACING_ARITHMETIC, B.SPACING_EXPONENTIAL)
    d.viscous_alpha, d.constant_viscosity = pick((1.0e-3, 0.0), (1.0e-2, 0.0), (0.0, 1.0e-5), (0.0, 1.0e-3), (0.0, 0.0))
    d.artificial_viscosity = pick(B.ARTVISC_NONE, B.ARTVISC_TW, B.ARTVISC_SN)
    d.artificial_viscosity_factor = pick(1.41, 2.0)
    d.artificial_viscosity_dissipation = int(rng.integers(2))
    d.heating_viscous = int(rng.integers(2))
    d.integrator = pick(B.INTEGRATOR_EULER, B.INTEGRATOR_LEAPFROG)
    d.flux_limiter = pick(B.LIMITER_VANLEER, B.LIMITER_VANLEER, B.LIMITER_MC)
    d.fast_transport = pick(1, 1, 0)
    d.omega_frame = pick(1.0, 0.0)
    d.cfl = pick(0.5, 0.4)
    d.radial_viscosity_factor = pick(1.0, 1.0, 2.0)
    for s in (0, 1):
        d.bc_sigma[s] = pick(B.BC_ZEROGRADIENT, B.BC_REFERENCE)
        d.bc_energy[s] = pick(B.BC_ZEROGRADIENT, B.BC_REFERENCE)
        d.bc_vrad[s] = pick(B.BC_ZEROGRADIENT, B.BC_REFERENCE, B.BC_REFLECTING, B.BC_OUTFLOW, B.BC_KEPLERIAN)
        d.bc_vaz[s] = pick(B.BC_ZEROGRADIENT, B.BC_REFERENCE, B.BC_KEPLERIAN, B.BC_ZEROSHEAR)
    d.damping = int(rng.integers(2))
    for arr in (d.damp_vrad, d.damp_vaz, d.damp_sigma, d.damp_energy):
        for s in (0, 1):
            arr[s] = pick(B.DAMP_NONE, B.DAMP_REFERENCE, B.DAMP_REFERENCE, B.DAMP_ZERO, B.DAMP_MEAN)
    for s in (0, 1):   # damping Sigma or e towards zero empties the zone within a few steps
        if d.damp_sigma[s] == B.DAMP_ZERO:
            d.damp_sigma[s] = B.DAMP_REFERENCE
        if d.damp_energy[s] == B.DAMP_ZERO:
            d.damp_energy[s] = B.DAMP_REFERENCE
    d.stabilize_viscosity = pick(0, 0, 0, 1, 2)
    if adiabatic:
        cooling = pick("none", "none", "surface_lin", "surface_const", "beta_zero", "beta_reference", "beta_floor")
        if cooling.startswith("surface"):
            d.cooling_surface = 1
            d.opacity = B.OPACITY_LIN if cooling == "surface_lin" else B.OPACITY_CONST
            d.kappa_const = 1.0e4
        elif cooling.startswith("beta"):
            d.cooling_beta, d.cooling_beta_value = 1, pick(5.0, 20.0)
            d.cooling_beta_reference = {"beta_zero": B.BETAREF_ZERO, "beta_reference": B.BETAREF_REFERENCE,
                                        "beta_floor": B.BETAREF_FLOOR}[cooling]
            d.cooling_beta_ramp_up = pick(0.0, 0.05)
    if rng.integers(4) == 0:    # ProfileCutoffOuter / Inner: the disk ends inside the grid
        d.profile_cutoff_outer, d.profile_cutoff_point_outer, d.profile_cutoff_width_outer = 1, pick(1.6, 2.0), pick(0.1, 0.2)
        if rng.integers(2):
            d.profile_cutoff_inner, d.profile_cutoff_point_inner, d.profile_cutoff_width_inner = 1, 0.6, pick(0.05, 0.1)
    nslabs = pick(1, 1, 2, 3)
    while nslabs > 1 and d.nr_global < 22 * nslabs:
        nslabs -= 1
    planet = bool(rng.integers(2))
    # appended draw (kept last so that earlier seeds keep their meaning): steps beyond the CFL time step (the
    # fused transport hands over to its fallback kernels past the shear limit)
    global _EXTRA
    _EXTRA = {"dt_scale": pick(1.0, 1.0, 1.0, 2.0, 3.5) if WIDE else 1.0}
    if WIDE:   # WriteMassFlow: the MASSFLOW grid is compared like a state grid
        d.write_massflow = int(rng.integers(3) == 0)
    if WIDE:   # initial-condition switches (init.cpp:255-343)
        d.initialize_pure_keplerian = int(rng.integers(4) == 0)
        d.initialize_vradial_zero = int(rng.integers(4) == 0)
        if rng.integers(6) == 0:
            d.set_sigma0, d.disk_mass = 1, pick(0.01, 0.002)
        if rng.integers(8) == 0:
            d.ic, d.disk_mass = B.IC_SPREADING_RING, 1.0e-3
    if WIDE and adiabatic and planet and rng.integers(3) == 0:
        # irradiation_single (SourceEuler.cpp:538-612): a hot star and a warm planet with a ramp-up time
        _EXTRA["irradiation"] = ([pick(4000.0, 10000.0) / setups.TEMP0_K, pick(0.0, 1500.0) / setups.TEMP0_K],
                                 [4.65e-3, 4.8e-4], [0.0, pick(0.0, 0.01)])
    # appended draw (round 3): BodyForceFromPotential: no -- CalculateAccelOnGas instead of the potential's gradient
    # (Pframeforce.cpp:96-189), in every source path (marching, per-loop) and integrator
    if rng.integers(5) == 0:
        d.body_force_from_potential = 0
    # appended draw (round 3): an explicit, ragged list of chunk lengths for the fused transport kernel (the table path
    # of k_transport_fused, which the built-in choice only takes on grids far larger than these)
    if rng.integers(3) == 0:
        _EXTRA["transport_chunks"] = [int(v) for v in rng.integers(1, 1 + max(2, d.nr_global // 2), size=int(rng.integers(1, 7)))]
    return d, nslabs, planet


GROWTH_CAP = 1.0e4   # widened bar at most 1e-13 x 1e4 = 1e-9; a more violent draw is compared over fewer steps instead
RELAXED = []         # (test, seed, steps, growth, tolerance): every draw that did not meet the plain 1e-10 bar
COMPARED = []        # every draw that was compared at all


def _tolerance(oracle_run, b, fields, worst):
    """1e-10, unless the draw is an unstable flow that amplifies rounding by orders of magnitude per step: then
    what the oracle does to cell-wise 1e-15 relative noise on its own input over the same steps (measured only
    when the plain bar is missed), never more than GROWTH_CAP."""
    if worst <= TOL:
        return TOL, 1.0
    noise = 1.0e-15
    b2 = oracle_run(noise)
    growth = max(rel_err(b2[k], b[k]) for k in fields) / noise
    # the two paths differ by a few 1e-14 before any amplification; one noise realisation
    return max(TOL, 1.0e-13 * min(growth, GROWTH_CAP)), growth


def _judge(test, seed, attempt):
    """attempt(nsteps) -> None (draw unusable) | (errs, dterr, tol, growth).  10 steps; a draw whose flow amplifies
    rounding beyond GROWTH_CAP in ten steps is compared after 3 instead, and skipped (and listed) if even that is
    beyond the cap -- never passed on a bar wider than 1e-9."""
    for nsteps in (10, 3):
        res = attempt(nsteps)
        if res is None:
            return
        errs, dterr, tol, growth = res
        if growth <= GROWTH_CAP:
            break
    else:
        RELAXED.append((test, seed, nsteps, growth, None))
        pytest.skip(f"seed {seed}: the oracle amplifies 1e-15 noise by {growth:.1e} in 3 steps: not a usable draw")
    COMPARED.append((test, seed))
    if tol > TOL:
        RELAXED.append((test, seed, nsteps, growth, tol))
        print(f"[fuzz] {test} seed {seed}: relaxed bar {tol:.1e} over {nsteps} steps (growth {growth:.1e}), "
              f"errors {max(errs.values()):.2e}, dt {dterr:.2e}")
    assert dterr <= 10 * tol, f"seed {seed}: time-step history differs by {dterr:.3e} (growth {growth:.1e}, {nsteps} steps)"
    for k, e in errs.items():
        assert e <= tol, f"seed {seed}: {k}: {e:.3e} (tolerance {tol:.1e}, growth {growth:.1e}, {nsteps} steps)"


@pytest.mark.parametrize("seed", range(FIRST, FIRST + NSEEDS))
def test_random_configuration(product, oracle, seed):
    d, nslabs, planet = draw(product, seed)
    bodies = setups.jupiter_bodies(d) if planet else None
    adiabatic = d.eos == B.EOS_IDEAL
    fields = ("sigma", "vrad", "vazi") + (("energy",) if adiabatic else ()) + (("massflow",) if d.write_massflow else ())

    def attempt(nsteps):
        try:
            b, dtb = run_pair(oracle, oracle, d, nsteps, bodies=bodies, nslabs=(1, 0), **_EXTRA)[0]
        except B.FcptError as err:   # a draw outside the supported space (e.g. NaN radii of a coarse exponential grid):
            assert "FCPT_EINVAL" in str(err)          # both libraries must refuse it the same way
            with pytest.raises(B.FcptError, match="FCPT_EINVAL"):
                run_pair(product, product, d, 1, bodies=bodies, nslabs=(1, 0))
            return None
        a, dta = run_pair(product, product, d, nsteps, bodies=bodies, nslabs=(nslabs, 0), **_EXTRA)[0]
        if not all(np.isfinite(b[k]).all() for k in b):
            pytest.skip("the oracle itself left the finite range: not a usable draw")
        if WIDE:   # every kernel reduces in a fixed order: a second run must give the same bits (a race would not)
            a2, dta2 = run_pair(product, product, d, nsteps, bodies=bodies, nslabs=(nslabs, 0), **_EXTRA)[0]
            assert dta2 == dta and all(np.array_equal(a2[k], a[k], equal_nan=True) for k in fields), f"seed {seed}: run-to-run difference"
        errs = {k: rel_err(a[k], b[k]) for k in fields}
        dterr = max(abs(x - y) / y for x, y in zip(dta, dtb))
        tol, growth = _tolerance(lambda noise: run_pair(oracle, oracle, d, nsteps, bodies=bodies, nslabs=(1, 0), noise=noise, **_EXTRA)[0][0],
                                 b, fields, max(max(errs.values()), 0.1 * dterr))
        return errs, dterr, tol, growth

    _judge("host_loop", seed, attempt)


def _device_loop(lib, d, bodies, nsteps, noise=0.0):
    from fargocpt_amd import driver
    dd = d.copy()
    dd.rank, dd.nranks = 0, 1
    radii = lib.radii(dd)
    from tests.util import perturb
    fields = perturb(lib.initial_fields(dd, radii), dd, 1e-3)
    if noise:
        rng = np.random.default_rng(7)
        fields = tuple(f * (1.0 + noise * rng.standard_normal(f.shape)) for f in fields)
    ctx = driver.make_context(lib, dd, fields=fields, radii=radii, bodies=bodies)
    S = driver.SlabSet([ctx])
    S.prepare()
    assert ctx.run_steps(nsteps) == nsteps
    out = S.gather()
    out["time"] = ctx.clock.time
    ctx.close()
    return out


@pytest.mark.parametrize("seed", range(FIRST, FIRST + NSEEDS, 2))
def test_random_configuration_device_loop(product, oracle, seed):
    """The same draws through fcpt_run_steps: dt from the CFL kernels to the policy kernel to the step without
    leaving the device, the fused transport with its fallback launches queued behind it and with the ring sums
    the source march leaves behind -- the path bench.py times."""
    d, _, planet = draw(product, seed)
    bodies = setups.jupiter_bodies(d) if planet else None
    fields = ("sigma", "vrad", "vazi") + (("energy",) if d.eos == B.EOS_IDEAL else ())

    def attempt(nsteps):
        nsteps = nsteps + 2
        try:
            b = _device_loop(oracle, d, bodies, nsteps)
        except B.FcptError as err:
            assert "FCPT_EINVAL" in str(err)
            return None
        if not all(np.isfinite(v).all() for v in b.values()):
            pytest.skip("the oracle itself left the finite range: not a usable draw")
        a = _device_loop(product, d, bodies, nsteps)
        if WIDE:
            a2 = _device_loop(product, d, bodies, nsteps)
            assert a2["time"] == a["time"] and all(np.array_equal(a2[k], a[k], equal_nan=True) for k in fields), f"seed {seed}: run-to-run difference"
        errs = {k: rel_err(a[k], b[k]) for k in fields}
        terr = abs(a["time"] - b["time"]) / abs(b["time"])
        tol, growth = _tolerance(lambda noise: _device_loop(oracle, d, bodies, nsteps, noise), b, fields, max(max(errs.values()), 0.1 * terr))
        return errs, terr, tol, growth

    _judge("device_loop", seed, attempt)


@pytest.mark.parametrize("seed", range(FIRST, FIRST + max(40, NSEEDS // 4)))
def test_random_slab_overlap_paths(product, seed, monkeypatch):
    """The two overlap devices of the N > 1 loop on random draws and slab positions: fcpt_cfl_begin (interior
    rings while the ghost rings travel) and fcpt_step_device_begin / _end (interior transport chunks on the side
    stream) must leave the bits of the plain sequence -- state, packed ghost rings, dt -- whatever the
    configuration, including those where they have to decline (leapfrog, narrow rings, damping outside the
    step kernels)."""
    import torch
    from fargocpt_amd import driver
    d, _, planet = draw(product, seed)
    rng = np.random.default_rng(77 + seed)
    nranks = int(rng.integers(2, 4))
    rank = int(rng.integers(0, nranks))
    d.nr_global = max(d.nr_global, 30) * nranks
    d.rank, d.nranks = rank, nranks
    radii = product.radii(d)
    if not np.isfinite(radii).all():
        pytest.skip("NaN radii: refused at create time (covered elsewhere)")
    from tests.util import perturb
    fields = perturb(product.initial_fields(d.copy(), radii), d, 1e-3)
    bodies = setups.jupiter_bodies(d) if planet else None
    got = []
    if seed % 2:   # half of the draws without the fallback kernels, so that the transport really splits
        monkeypatch.setenv("FCPT_TRANSPORT_FALLBACK", "0")
    for overlap in (True, False):
        monkeypatch.setenv("FCPT_CFL_SPLIT", "1" if overlap else "0")
        ctx = driver.make_context(product, d, fields=fields, radii=radii, bodies=bodies)
        cnt = ctx.exchange_count()
        bufs = [torch.zeros(cnt, dtype=torch.float64, device="cuda") if has else None
                for has in (rank > 0, rank < nranks - 1)]
        ptr = lambda b: None if b is None else b.data_ptr()
        dt_dev = torch.zeros(1, dtype=torch.float64, device="cuda")
        for _ in range(2):
            ctx.calculate_timestep(ctx.cfl())
        hist, packed = [], []
        for n in range(6):
            ctx.cfl_device(dt_dev.data_ptr())
            ctx.calculate_timestep_device(dt_dev.data_ptr())
            if overlap:
                ctx.step_device_begin()
                ctx.exchange_pack(ptr(bufs[0]), ptr(bufs[1]))
                ctx.step_device_end()
                ctx.cfl_begin()
            else:
                ctx.step_device()
                ctx.exchange_pack(ptr(bufs[0]), ptr(bufs[1]))
            ctx.synchronize()
            packed.append([None if b is None else b.cpu().numpy().copy() for b in bufs])
            for b in bufs:                  # stand-in for the neighbours: the ghost rows get new values
                if b is not None:
                    b.mul_(1.0 + 1.0e-6)
            torch.cuda.synchronize()
            ctx.exchange_unpack(ptr(bufs[0]), ptr(bufs[1]))
            ctx.post_device()
            hist.append(float(dt_dev.cpu()[0]))
        try:
            st = ctx.state()
        except B.FcptError as err:
            ctx.close()
            assert "FCPT_ESHEAR" in str(err) and seed % 2
            pytest.skip("beyond the one-lane shift without the fallback kernels")
        st["time"] = ctx.clock.time
        got.append((st, hist, packed))
        ctx.close()
    assert got[0][1] == got[1][1] or (np.isnan(got[0][1]).any() and np.isnan(got[1][1]).any())
    for k, v in got[0][0].items():
        assert np.array_equal(v, got[1][0][k], equal_nan=True), k
    for pa, pb in zip(got[0][2], got[1][2]):
        for x, y in zip(pa, pb):
            assert (x is None and y is None) or np.array_equal(x, y, equal_nan=True)


def test_zz_share_of_relaxed_draws():
    """Runs last: the draws that did not meet the plain 1e-10 bar stay a small, listed minority (each of them was
    held to at most 1e-9, over ten or three steps)."""
    import json
    import os
    if not COMPARED:
        pytest.skip("no draws compared in this session")
    rec = {"compared": len(COMPARED), "relaxed": [dict(test=t, seed=sd, steps=n, growth=g, tolerance=tol)
                                                   for t, sd, n, g, tol in RELAXED], "growth_cap": GROWTH_CAP}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "fuzz_relaxed.json"), "w") as f:
            json.dump(rec, f, indent=1)
    print(f"[fuzz] {len(RELAXED)} of {len(COMPARED)} compared draws needed a relaxed bar or fewer steps: "
          f"{[(t, sd) for t, sd, *_ in RELAXED]}")
    assert len(RELAXED) <= max(2, 0.15 * len(COMPARED)), rec
