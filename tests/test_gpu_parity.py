"""Parity of the HIP path against the CPU oracle on identical inputs (through the C ABI).

Tolerance: max|a-b|/max|b| <= 1e-10 on Sigma, v_r, v_phi (and e) after N steps -- the
bar of BASELINE.json's north_star.  Observed differences are ~1e-14 (FMA contraction,
tree vs serial ring sums, OCML vs glibc exp)."""
import numpy as np
import pytest

from fargocpt_amd import binding as B, setups
from tests.util import rel_err, run_pair

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _check(outs, fields, tol=TOL):
    (a, dta), (b, dtb) = outs
    assert np.allclose(dta, dtb, rtol=1e-9, atol=0), "time-step history differs"
    for k in fields:
        e = rel_err(a[k], b[k])
        assert e <= tol, f"{k}: {e:.3e} > {tol}"


def test_spreading_ring_128x384(product, oracle):
    """BASELINE config 1: spreading ring, locally isothermal, 128x384, 100 steps."""
    d = setups.spreading_ring(product, 128, 384)
    d.first_dt = 1e-3
    _check(run_pair(product, oracle, d, 100), ("sigma", "vrad", "vazi"))


def test_iso_planet_128x384(product, oracle):
    """BASELINE config 2 physics (examples/config.yml) at the reference's own size."""
    d = setups.planet_disk(product, 128, 384)
    _check(run_pair(product, oracle, d, 60, bodies=setups.jupiter_bodies(d)), ("sigma", "vrad", "vazi"))


def test_adiabatic_alpha_96x288(product, oracle):
    """BASELINE config 3 physics: ideal EOS + alpha viscosity + viscous heating."""
    d = setups.planet_disk(product, 96, 288, adiabatic=True)
    _check(run_pair(product, oracle, d, 40), ("sigma", "vrad", "vazi", "energy"))


@pytest.mark.parametrize("av", ["SN", "TW"])
def test_shocktube_4096x4(product, oracle, av):
    """BASELINE config 5: shock tube 4096x4 (van Leer limiter correctness).  Rounding
    differences are amplified by the steepening shock (1e-15 after one step, ~1e-12 after
    20), so the bar is the north-star 1e-10, checked after the shock has formed."""
    d = setups.shocktube(product, 4096, 4, av)
    d.first_dt = 1e-6
    _check(run_pair(product, oracle, d, 60, amp=0.0), ("sigma", "vrad", "vazi", "energy"))


def test_shocktube_to_monitor_time(product, oracle):
    """Run the reference's shock-tube setup to its snapshot time with snapping."""
    d = setups.shocktube(product, 100, 2, "SN")
    (a, dta), (b, dtb) = run_pair(product, oracle, d, 270, amp=0.0, snap=True)
    assert abs(sum(dta) - 0.228) < 1e-12 and abs(sum(dtb) - 0.228) < 1e-12
    for k in ("sigma", "vrad", "energy"):
        assert rel_err(a[k], b[k]) <= 1e-11


def test_two_slabs_match_one(product, oracle):
    """Radial split (split.cpp) + ghost exchange (commbound.cpp): 2 HIP slabs vs 1 oracle slab.
    The reference itself agrees to 4e-13 between 1 and 2 ranks."""
    d = setups.planet_disk(product, 64, 192)
    _check(run_pair(product, oracle, d, 30, nslabs=(2, 1)), ("sigma", "vrad", "vazi"))
    d = setups.planet_disk(product, 96, 64, adiabatic=True)
    _check(run_pair(product, oracle, d, 20, nslabs=(3, 1)), ("sigma", "vrad", "vazi", "energy"))


def test_marching_source_kernel_variants(product, oracle):
    """The one-pass wave-marching source kernel (isothermal, Nphi >= 128) for SN and no
    artificial viscosity, constant viscosity, and an odd ring count / Nphi not a multiple of 59."""
    d = setups.planet_disk(product, 45, 250)
    d.artificial_viscosity = B.ARTVISC_SN
    _check(run_pair(product, oracle, d, 25, bodies=setups.jupiter_bodies(d)), ("sigma", "vrad", "vazi"))
    d = setups.planet_disk(product, 40, 128)
    d.artificial_viscosity = B.ARTVISC_NONE
    d.viscous_alpha, d.constant_viscosity = 0.0, 1e-5
    _check(run_pair(product, oracle, d, 25), ("sigma", "vrad", "vazi"))


def test_tiled_azimuthal_sweep_32x640(product, oracle):
    """Nphi > 384 takes the tiled mode of the fused azimuthal kernel (segments of 256 cells)."""
    d = setups.planet_disk(product, 32, 640)
    _check(run_pair(product, oracle, d, 25, bodies=setups.jupiter_bodies(d)), ("sigma", "vrad", "vazi"))
    d = setups.planet_disk(product, 32, 640, adiabatic=True)
    _check(run_pair(product, oracle, d, 25), ("sigma", "vrad", "vazi", "energy"))


def test_unfused_paths_agree(product, oracle, monkeypatch):
    """The per-loop kernels (FCPT_FUSED_SOURCE=0, FCPT_THETA_FUSED=0) stay available as a
    cross-check of the fused ones."""
    monkeypatch.setenv("FCPT_FUSED_SOURCE", "0")
    monkeypatch.setenv("FCPT_THETA_FUSED", "0")
    d = setups.planet_disk(product, 48, 96, adiabatic=True)
    _check(run_pair(product, oracle, d, 20), ("sigma", "vrad", "vazi", "energy"))


def test_mc_limiter_and_standard_transport(product, oracle):
    d = setups.planet_disk(product, 48, 64)
    d.flux_limiter = B.LIMITER_MC
    d.fast_transport = 0
    _check(run_pair(product, oracle, d, 20), ("sigma", "vrad", "vazi"))


def test_device_resident_dt_loop(product):
    """fcpt_run_steps (snap=0) and the DistributedSlab.step_async loop keep dt on the device;
    both must reproduce the host-driven loop (cfl -> calculate_timestep -> step -> post) exactly."""
    import torch
    from fargocpt_amd import driver
    from fargocpt_amd.parallel import DistributedSlab
    d = setups.planet_disk(product, 64, 256)
    radii = product.radii(d)
    fields = product.initial_fields(d.copy(), radii)
    states = []
    for mode in ("host", "run_steps", "async"):
        ctx = driver.make_context(product, d, fields=fields, radii=radii, bodies=setups.jupiter_bodies(d))
        S = driver.SlabSet([ctx])
        S.prepare()
        if mode == "host":
            S.run(12)
        elif mode == "run_steps":
            assert ctx.run_steps(12) == 12
        else:
            slab = DistributedSlab(ctx, device=torch.device("cuda", 0))
            for _ in range(12):
                slab.step_async()
        st = ctx.state()
        st["time"] = ctx.clock.time
        states.append(st)
        ctx.close()
    for other in states[1:]:
        assert other["time"] == states[0]["time"]
        for k in ("sigma", "vrad", "vazi"):
            assert np.array_equal(other[k], states[0][k]), k


def test_weak_scaling_grid_four_slabs(product, oracle):
    """bench.py's N > 1 configuration at reduced size: the log grid is extended outward
    (rmax = rmin * 6.25^N) and split into N radial slabs; 4 HIP slabs vs 1 oracle slab."""
    n = 4
    d = setups.planet_disk(product, 4 * 40, 160)
    d.rmax = d.rmin * (2.5 / 0.4) ** n
    d.damping_time_radius_outer = d.rmax
    _check(run_pair(product, oracle, d, 15, nslabs=(n, 1), bodies=setups.jupiter_bodies(d)),
           ("sigma", "vrad", "vazi"))


@pytest.mark.parametrize("adiabatic,nphi", [(False, 256), (True, 96), (False, 64), (True, 320)])
def test_leapfrog_integrator(product, oracle, adiabatic, nphi):
    """step_LeapFrog (src/simulation.cpp:276-459): kick 1/2, drift, kick 2/2 with the mid-step
    bodies; isothermal (marching source kernel and the narrow-grid fused path) and adiabatic."""
    d = setups.planet_disk(product, 48, nphi, adiabatic=adiabatic)
    d.integrator = B.INTEGRATOR_LEAPFROG
    fields = ("sigma", "vrad", "vazi") + (("energy",) if adiabatic else ())
    _check(run_pair(product, oracle, d, 25, bodies=setups.jupiter_bodies(d)), fields)
    d = setups.shocktube(product, 512, 4, "TW", leapfrog=True)
    d.first_dt = 1e-6
    _check(run_pair(product, oracle, d, 40, amp=0.0), ("sigma", "vrad", "vazi", "energy"))


@pytest.mark.parametrize("case", ["outflow_zeroshear", "reference_bc", "keplerian_vrad", "damp_zero", "damp_mean",
                                  "exponential_grid", "arithmetic_grid", "odd_nphi", "no_damping_tiny"])
def test_boundary_damping_grid_variants(product, oracle, case):
    """Per-variable boundary conditions (src/boundary_conditions/*.cpp), damping targets
    (damping.cpp:311-700), the three grid spacings (init.cpp:92-145) and ragged sizes."""
    d = setups.planet_disk(product, 40, 144)
    if case == "outflow_zeroshear":
        for s in (0, 1):
            d.bc_vrad[s], d.bc_vaz[s] = B.BC_OUTFLOW, B.BC_ZEROSHEAR
    elif case == "reference_bc":
        for s in (0, 1):
            d.bc_sigma[s] = d.bc_energy[s] = d.bc_vrad[s] = d.bc_vaz[s] = B.BC_REFERENCE
    elif case == "keplerian_vrad":
        for s in (0, 1):
            d.bc_vrad[s], d.bc_vaz[s] = B.BC_KEPLERIAN, B.BC_ZEROGRADIENT
    elif case == "damp_zero":
        for arr in (d.damp_vrad, d.damp_sigma):
            arr[0] = arr[1] = B.DAMP_ZERO
    elif case == "damp_mean":
        for arr in (d.damp_vrad, d.damp_vaz, d.damp_sigma):
            arr[0] = arr[1] = B.DAMP_MEAN
    elif case == "exponential_grid":
        d.radial_spacing = B.SPACING_EXPONENTIAL
    elif case == "arithmetic_grid":
        d.radial_spacing = B.SPACING_ARITHMETIC
    elif case == "odd_nphi":
        d.nr_global, d.nphi = 37, 131
    elif case == "no_damping_tiny":
        d.nr_global, d.nphi, d.damping = 16, 8, 0
    _check(run_pair(product, oracle, d, 20, bodies=setups.jupiter_bodies(d)), ("sigma", "vrad", "vazi"))
    da = d.copy()
    da.eos = B.EOS_IDEAL
    _check(run_pair(product, oracle, da, 12), ("sigma", "vrad", "vazi", "energy"))


FUSED_CASES = ["iso_tw", "iso_sn_mc", "iso_noav_constnu_standard", "adiabatic", "adiabatic_sn", "outflow_zeroshear",
               "reference_bc_damp_zero", "no_damping", "arithmetic_odd", "three_slabs", "leapfrog", "two_cells_per_lane"]


@pytest.mark.parametrize("case", FUSED_CASES)
def test_fused_transport_variants(product, oracle, case, monkeypatch):
    """k_transport_fused (radial + azimuthal transport + velocities in one marching kernel) takes
    rings of Nphi >= 256: every physics / boundary / grid variant of the path at such sizes."""
    d = setups.planet_disk(product, 44, 320, adiabatic=case.startswith("adiabatic"))
    fields = ("sigma", "vrad", "vazi") + (("energy",) if case.startswith("adiabatic") else ())
    kw = {}
    if case == "iso_sn_mc":
        d.artificial_viscosity, d.flux_limiter = B.ARTVISC_SN, B.LIMITER_MC
    elif case == "iso_noav_constnu_standard":
        d.artificial_viscosity, d.fast_transport = B.ARTVISC_NONE, 0
        d.viscous_alpha, d.constant_viscosity = 0.0, 1e-5
    elif case == "adiabatic_sn":
        d.artificial_viscosity = B.ARTVISC_SN
    elif case == "outflow_zeroshear":
        for s in (0, 1):
            d.bc_vrad[s], d.bc_vaz[s] = B.BC_OUTFLOW, B.BC_ZEROSHEAR
    elif case == "reference_bc_damp_zero":
        for s in (0, 1):
            d.bc_sigma[s] = d.bc_energy[s] = d.bc_vrad[s] = d.bc_vaz[s] = B.BC_REFERENCE
        for arr in (d.damp_vrad, d.damp_sigma):
            arr[0] = arr[1] = B.DAMP_ZERO
    elif case == "no_damping":
        d.damping = 0
    elif case == "arithmetic_odd":
        d.radial_spacing = B.SPACING_ARITHMETIC
        d.nr_global, d.nphi = 37, 263
    elif case == "three_slabs":
        d.nr_global = 60
        kw["nslabs"] = (3, 1)
    elif case == "leapfrog":
        d.integrator = B.INTEGRATOR_LEAPFROG
    elif case == "two_cells_per_lane":
        monkeypatch.setenv("FCPT_TRANSPORT_FUSED", "2")
    _check(run_pair(product, oracle, d, 25, bodies=setups.jupiter_bodies(d), **kw), fields)


def test_transport_beyond_shear_limit_falls_back(product, oracle, monkeypatch):
    """k_transport_fused couples rings i-1 and i through one lane shift, which covers
    |Nshift[i] - Nshift[i-1]| <= 1 (every dt inside the FARGO shear limit, cfl.cpp:207-220).  A step
    of 4x the CFL time step breaks that: the kernel must give up and the unfused kernels queued
    behind it must produce the step.  (They are queued behind every fused launch: the CFL policy's dt does
    not rule the situation out when the source step itself changes v_phi violently.)"""
    d = setups.planet_disk(product, 48, 512)
    d.damping = 0
    d.first_dt = 1.0  # no 1.1x ramp: the first step already runs at the CFL limit
    bodies = setups.jupiter_bodies(d)
    _check(run_pair(product, oracle, d, 3, bodies=bodies, dt_scale=4.0), ("sigma", "vrad", "vazi"))
    # without the fallback kernels the situation is detected and reported, not computed wrongly
    monkeypatch.setenv("FCPT_TRANSPORT_FALLBACK", "0")
    with pytest.raises(B.FcptError, match="FCPT_ESHEAR"):
        run_pair(product, oracle, d, 3, bodies=bodies, dt_scale=4.0)


def test_fallback_in_three_slabs_and_back_to_the_fused_kernel(product, oracle):
    """Round 3: the shift-jump flag is a sequence stamp left by k_ring_mean (nothing resets it), the fused launch runs
    the radial sweep itself when it is raised and one gated azimuthal launch finishes the step -- no grid barrier.
    (a) three slabs of one process, energy equation, steps of 4x the CFL step: each slab decides for itself (the
    shear is strongest in the inner slab), all paths must give the oracle's single-slab result; (b) a context that
    has just taken the fallback path takes the fused kernel again in the next, ordinary step (a stale stamp would
    keep it on the two-kernel path for ever): the read-only option transport_fell_back reports the stamps."""
    from fargocpt_amd import driver
    d = setups.planet_disk(product, 96, 512, adiabatic=True)
    d.damping = 0
    d.first_dt = 1.0
    _check(run_pair(product, oracle, d, 3, bodies=setups.jupiter_bodies(d), dt_scale=4.0, nslabs=(3, 1)),
           ("sigma", "vrad", "vazi", "energy"))
    d = setups.planet_disk(product, 48, 512)   # (coarse in radius: the shear between neighbouring rings is what limits dt)
    d.damping = 0
    d.first_dt = 1.0
    ctx = driver.make_context(product, d, bodies=setups.jupiter_bodies(d))
    S = driver.SlabSet([ctx])
    S.prepare()
    seen = []
    for scale in (1.0, 4.0, 4.0, 1.0, 1.0):
        S.dt_scale = scale
        S.run(1)
        seen.append(ctx.get_option("transport_fell_back"))
    ctx.close()
    # (whether the second long step exceeds the limit again depends on the state the first one left)
    assert seen[0] == 0 and seen[1] == 1 and seen[3:] == [0, 0], seen


@pytest.mark.parametrize("adiabatic", [False, True])
def test_disk_on_body_accel(product, oracle, adiabatic):
    """ComputeDiskOnPlanetAccel (Force.cpp:23-122): the four sums (inner /
    outer rings, x / y) of the gas's specific force on the planet, after a few steps have built the
    wake.  The sums cancel to ~1e-3 of their terms in a nearly axisymmetric disk, so the bar is set
    against the sum of the terms' magnitudes (G M_disk / d^2 scale), not against the net value."""
    from fargocpt_amd import driver
    d = setups.planet_disk(product, 48, 256, adiabatic=adiabatic)
    bodies = setups.jupiter_bodies(d)
    res = []
    for L in (product, oracle):
        ctx = driver.make_context(L, d, bodies=bodies)
        S = driver.SlabSet([ctx])
        S.prepare()
        S.run(8)
        hill = (bodies[2][1] / 3.0) ** (1.0 / 3.0)
        a = [ctx.disk_on_body_accel(1.0, 0.0, 1.0),                           # H-based smoothing per cell
             ctx.disk_on_body_accel(1.0, 0.0, 1.0, 0.6 * 0.05, 0.5 * hill),   # planet-location + cubic smoothing
             ctx.disk_on_body_accel(0.0, 0.0, 0.0, 0.0, 0.0)]                 # the star, no smoothing
        rng = np.random.default_rng(11)   # bodies anywhere: inside the inner hole, between rings, beyond the disk
        for _ in range(24):
            r, ph = rng.uniform(0.05, 3.5), rng.uniform(0.0, 2 * np.pi)
            fixed = float(rng.choice([-1.0, 0.0, 0.03]))
            a.append(ctx.disk_on_body_accel(r * np.cos(ph), r * np.sin(ph), r, fixed, float(rng.choice([0.0, 0.4 * hill]))))
        sig = ctx.download(B.F_SIGMA)
        res.append((a, sig))
        ctx.close()
    (a, sig), (b, _) = res
    scale = float(np.abs(sig).sum()) * 2 * np.pi * 2.5 ** 2 / sig.size  # ~ G M_disk with G = 1, d ~ 1
    for x, y in zip(a, b):
        assert np.all(np.abs(x - y) <= 1e-10 * np.abs(y).max() + 1e-12 * scale), (x, y)
        assert np.abs(y).max() > 0


@pytest.mark.parametrize("case", ["surface_lin_march", "surface_simple_beta_reference_narrow", "beta_model_ramp_leapfrog",
                                  "surface_const_beta_floor"])
def test_cooling_terms(product, oracle, case):
    """calculate_qminus in SubStep3 (SourceEuler.cpp:632-820,931-950): thermal
    surface cooling with the Lin / Const / Simple opacities and beta cooling towards zero / the reference
    state / the model profile / the floor, in the marching kernel (Nphi >= 128) and the per-cell kernels."""
    nphi = 96 if "narrow" in case else 320
    d = setups.planet_disk(product, 44, nphi, adiabatic=True)
    d.heating_cooling_cfl_limit = 10.0
    if case == "surface_lin_march":
        d.cooling_surface, d.opacity = 1, B.OPACITY_LIN
    elif case == "surface_simple_beta_reference_narrow":
        d.cooling_surface, d.opacity, d.kappa_const = 1, B.OPACITY_SIMPLE, 17.77
        d.tau_factor, d.density_factor = 1.0, 2.0
        d.cooling_beta, d.cooling_beta_value, d.cooling_beta_reference = 1, 10.0, B.BETAREF_REFERENCE
    elif case == "beta_model_ramp_leapfrog":
        d.cooling_beta, d.cooling_beta_value, d.cooling_beta_reference = 1, 5.0, B.BETAREF_MODEL
        d.cooling_beta_ramp_up = 0.05
        d.integrator = B.INTEGRATOR_LEAPFROG
    else:
        d.cooling_surface, d.opacity, d.kappa_const = 1, B.OPACITY_CONST, 1.0e4
        d.cooling_beta, d.cooling_beta_value, d.cooling_beta_reference = 1, 20.0, B.BETAREF_FLOOR
    _check(run_pair(product, oracle, d, 25, bodies=setups.jupiter_bodies(d)), ("sigma", "vrad", "vazi", "energy"))


def test_temperature_test_setup_parity(product, oracle):
    """test/TemperatureTest/angelo.yml (100 x 2, leapfrog, viscous heating against thermal cooling with
    kappa ~ T^2): 400 steps of the HIP path against the oracle, which passes the reference's own
    criterion on this setup (tests/test_oracle_known_answers.py)."""
    d = setups.temperature_test(product)
    _check(run_pair(product, oracle, d, 400, amp=0.0), ("sigma", "vrad", "vazi", "energy"))


def test_irradiation_parity(product, oracle):
    """irradiation_single (SourceEuler.cpp:538-612) in SubStep3: the reference's irradiation setup (200 x 2,
    per-cell kernels) for 300 steps, and a planet disk at 44 x 320 (marching kernel) heated by the star and
    by a hot planet with a ramp-up time."""
    from fargocpt_amd import driver
    d, bodies, irr = setups.irradiation_test(product)
    outs = []
    for L in (product, oracle):
        ctx = driver.make_context(L, d, bodies=bodies, irradiation=irr)
        S = driver.SlabSet([ctx])
        S.prepare()
        dts = S.run(300)
        outs.append((S.gather(), dts))
        ctx.close()
    _check(outs, ("sigma", "vrad", "vazi", "energy"))
    d = setups.planet_disk(product, 44, 320, adiabatic=True)
    d.cooling_surface, d.opacity = 1, B.OPACITY_LIN
    x, y, m = setups.jupiter_bodies(d)
    irr = ([5800.0 / setups.TEMP0_K, 1500.0 / setups.TEMP0_K], [4.65e-3, 4.7e-4], [0.0, 0.05])
    outs = []
    for L in (product, oracle):
        ctx = driver.make_context(L, d, bodies=(x, y, m, [0.0, 0.05]), irradiation=irr)
        S = driver.SlabSet([ctx])
        S.prepare()
        dts = S.run(25)
        outs.append((S.gather(), dts))
        ctx.close()
    _check(outs, ("sigma", "vrad", "vazi", "energy"))


@pytest.mark.parametrize("case", ["update_iso", "update_adiabatic_two_slabs", "dt_leapfrog", "dt_radial_factor",
                                  "dt_first_step_adiabatic"])
def test_stabilize_viscosity(product, oracle, case):
    """StabilizeViscosity 1 (viscosity.cpp:386-391|413-417: the viscous velocity update is damped where
    dt c < -1, forced here by stepping with 3x the CFL step) and 2 (cfl.cpp:331-351: dt <= -CFL / c, binding
    here through the leapfrog's 0.6 on the kinematic limit or a radial viscosity factor): fields, dt history
    and the correction factors themselves against the oracle, and the switch must change the run."""
    from fargocpt_amd import driver
    mode = 1 if case.startswith("update") else 2
    adiabatic = "adiabatic" in case
    d = setups.planet_disk(product, 48, 192, adiabatic=adiabatic)
    d.viscous_alpha, d.constant_viscosity = 0.0, 1.0e-2
    if case == "dt_first_step_adiabatic":
        # long rings: the limit binds from the first step on, through the factors that init_euler's
        # compute_heating_cooling_for_CFL leaves behind (SourceEuler.cpp:284,1507-1513; found by the fuzzer)
        d = setups.planet_disk(product, 40, 900, adiabatic=True)
        d.viscous_alpha, d.constant_viscosity = 0.0, 1.0e-3
        d.integrator, d.radial_viscosity_factor = B.INTEGRATOR_LEAPFROG, 2.0
    d.stabilize_viscosity = mode
    if case == "dt_leapfrog":
        d.integrator = B.INTEGRATOR_LEAPFROG
    if case == "dt_radial_factor":
        d.radial_viscosity_factor = 2.5
    scale = 3.0 if mode == 1 else 1.0
    fields = ("sigma", "vrad", "vazi", "energy") if adiabatic else ("sigma", "vrad", "vazi")
    outs = run_pair(product, oracle, d, 70, nslabs=(2, 1) if "two_slabs" in case else (1, 1), dt_scale=scale)
    _check(outs, fields)
    d0 = d.copy()
    d0.stabilize_viscosity = 0
    plain, dt_plain = run_pair(product, product, d0, 70, nslabs=(1, 0), dt_scale=scale)[0]
    if mode == 2:
        assert outs[0][1][-1] < 0.95 * dt_plain[-1], "the stability limit did not bind"
    if case == "dt_first_step_adiabatic":
        assert outs[0][1][0] < 0.95 * dt_plain[0], "the stability limit did not bind at the first step"
    else:
        assert rel_err(outs[0][0]["vrad"], plain["vrad"]) > 1e-3, "the update was never damped"
    # the factors (t_data VISCOSITY_CORRECTION_FACTOR_PHI|R) after one more kick
    facs = []
    for L in (product, oracle):
        dd = d.copy()
        dd.rank, dd.nranks = 0, 1
        ctx = driver.make_context(L, dd)
        S = driver.SlabSet([ctx])
        S.prepare()
        S.run(3)
        facs.append((ctx.download(B.F_VISC_CFAC_PHI), ctx.download(B.F_VISC_CFAC_R)))
        ctx.close()
    assert rel_err(facs[0][0], facs[1][0]) <= TOL and rel_err(facs[0][1], facs[1][1]) <= TOL
    assert (facs[0][0][1:] < 0).all() and (facs[0][1][1:] < 0).all()
    with pytest.raises(B.FcptError):
        ctx = driver.make_context(product, d0)
        try:
            ctx.download(B.F_VISC_CFAC_R)   # only with StabilizeViscosity
        finally:
            ctx.close()


@pytest.mark.parametrize("adiabatic,rank,nranks", [(False, 1, 3), (True, 1, 3), (False, 0, 2), (True, 1, 2)])
def test_cfl_split_around_the_ghost_exchange(product, adiabatic, rank, nranks, monkeypatch):
    """fcpt_cfl_begin (interior rings, queued while the ghost rings travel) + fcpt_cfl (the rest) must give the
    dt of the unsplit reduction, bit for bit, on slabs with neighbours (middle slab; first and last slab, whose
    damping zones are handled inside the step kernels) -- and must really split (two launches of the ring
    kernel instead of one)."""
    import torch
    from fargocpt_amd import driver
    d = setups.planet_disk(product, nranks * 40, 320, adiabatic=adiabatic)
    d.rank, d.nranks = rank, nranks
    radii = product.radii(d)
    fields = product.initial_fields(d.copy(), radii)
    names = product.kernel_names()
    got = []
    for split in ("1", "0"):
        monkeypatch.setenv("FCPT_CFL_SPLIT", split)
        ctx = driver.make_context(product, d, fields=fields, radii=radii, bodies=setups.jupiter_bodies(d))
        cnt = ctx.exchange_count()
        bufs = [torch.zeros(cnt, dtype=torch.float64, device="cuda") if has else None
                for has in (rank > 0, rank < nranks - 1)]
        ptr = lambda b: None if b is None else b.data_ptr()
        dts, launches = [], 0
        for _ in range(2):
            ctx.calculate_timestep(ctx.cfl())
        for n in range(6):
            dt = ctx.calculate_timestep(ctx.cfl() if n == 0 else dts[-1])
            ctx.step(dt)
            ctx.exchange_pack(ptr(bufs[0]), ptr(bufs[1]))
            ctx.synchronize()
            for b in bufs:                  # stand-in for the neighbours: the ghost rows get new values
                if b is not None:
                    b.mul_(1.0 + 1.0e-4)
            torch.cuda.synchronize()
            ctx.profile_start([names.index("k_cfl_rings")], max_launches=8)
            ctx.cfl_begin()
            ctx.exchange_unpack(ptr(bufs[0]), ptr(bufs[1]))
            ctx.post(dt)
            dts.append(ctx.cfl())
            launches += ctx.profile_stop()["k_cfl_rings"][1]
        got.append((dts, launches, ctx.state()))
        ctx.close()
    assert got[0][0] == got[1][0]
    assert got[0][1] == 12 and got[1][1] == 6
    for k in ("sigma", "vrad", "vazi"):
        assert np.array_equal(got[0][2][k], got[1][2][k])


@pytest.mark.parametrize("adiabatic", [False, True])
def test_wide_rings_6144(product, oracle, adiabatic):
    """BASELINE config 4's ring length (Nphi = 6144 > 4096): the one-block-per-ring CFL kernel with 16 cell
    pairs per thread, the fused transport kernel over 96 column tiles."""
    d = setups.planet_disk(product, 24, 6144, adiabatic=adiabatic)
    fields = ("sigma", "vrad", "vazi", "energy") if adiabatic else ("sigma", "vrad", "vazi")
    _check(run_pair(product, oracle, d, 12, bodies=setups.jupiter_bodies(d)), fields)


@pytest.mark.parametrize("adiabatic,rank,nranks", [(False, 1, 3), (True, 1, 3), (False, 0, 2), (False, 1, 2)])
def test_step_split_around_the_ghost_exchange(product, adiabatic, rank, nranks, monkeypatch):
    """fcpt_step_device_begin / _end: the transport chunks with the neighbours' ghost rings on the caller's
    stream, the others on the library's side stream under the pack kernel.  Same bits as fcpt_step_device, the
    packed rings included, and really two launches of the marching kernel per step."""
    import torch
    from fargocpt_amd import driver
    monkeypatch.setenv("FCPT_TRANSPORT_FALLBACK", "0")   # the split needs the fused kernel alone (benign flow here)
    d = setups.planet_disk(product, nranks * 120, 320, adiabatic=adiabatic)
    d.rank, d.nranks = rank, nranks
    radii = product.radii(d)
    fields = product.initial_fields(d.copy(), radii)
    names = product.kernel_names()
    got = []
    for split in (True, False):
        ctx = driver.make_context(product, d, fields=fields, radii=radii, bodies=setups.jupiter_bodies(d))
        cnt = ctx.exchange_count()
        bufs = [torch.zeros(cnt, dtype=torch.float64, device="cuda") if has else None
                for has in (rank > 0, rank < nranks - 1)]
        ptr = lambda b: None if b is None else b.data_ptr()
        dt_dev = torch.zeros(1, dtype=torch.float64, device="cuda")
        for _ in range(2):
            ctx.calculate_timestep(ctx.cfl())
        launches, packed = 0, []
        for n in range(6):
            ctx.cfl_device(dt_dev.data_ptr())
            ctx.calculate_timestep_device(dt_dev.data_ptr())
            ctx.profile_start([names.index("k_transport_fused")], max_launches=8)
            if split:
                ctx.step_device_begin()
                ctx.exchange_pack(ptr(bufs[0]), ptr(bufs[1]))
                ctx.step_device_end()
            else:
                ctx.step_device()
                ctx.exchange_pack(ptr(bufs[0]), ptr(bufs[1]))
            launches += ctx.profile_stop()["k_transport_fused"][1]
            ctx.synchronize()
            packed.append([None if b is None else b.cpu().numpy().copy() for b in bufs])
            for b in bufs:                  # stand-in for the neighbours: the ghost rows get new values
                if b is not None:
                    b.mul_(1.0 + 1.0e-4)
            torch.cuda.synchronize()
            ctx.exchange_unpack(ptr(bufs[0]), ptr(bufs[1]))
            ctx.post_device()
        st = ctx.state()
        st["time"] = ctx.clock.time
        got.append((st, launches, packed))
        ctx.close()
    assert got[0][1] == 12 and got[1][1] == 6
    assert got[0][0]["time"] == got[1][0]["time"]
    for k in ("sigma", "vrad", "vazi") + (("energy",) if adiabatic else ()):
        assert np.array_equal(got[0][0][k], got[1][0][k]), k
    for pa, pb in zip(got[0][2], got[1][2]):
        for x, y in zip(pa, pb):
            assert (x is None and y is None) or np.array_equal(x, y)


@pytest.mark.parametrize("adiabatic,leapfrog,nphi", [(False, False, 320), (False, True, 320), (True, True, 288), (True, False, 96)])
def test_moving_bodies_indirect_term_and_midstep_positions(product, oracle, adiabatic, leapfrog, nphi):
    """What the host loop does around every step when the frame does not corotate: fcpt_set_bodies with the
    bodies' new positions, cubic smoothing radii and the indirect term (Pframeforce.cpp:21-189), for leapfrog
    also fcpt_set_bodies_midstep with the positions at t + dt/2 (simulation.cpp:359-366).  Two planets on
    circular orbits, OmegaFrame = 0."""
    from fargocpt_amd import driver
    d = setups.planet_disk(product, 56, nphi, adiabatic=adiabatic)
    d.omega_frame = 0.0
    if leapfrog:
        d.integrator = B.INTEGRATOR_LEAPFROG
    m_star, planets = d.hydro_center_mass, [(1.0, 1.0e-3, 0.0), (1.6, 3.0e-4, 2.0)]   # (a, mass, phase)

    def bodies_at(t):
        x, y, m, rsm = [0.0], [0.0], [m_star], [0.0]
        ix = iy = 0.0
        for a, mp, ph in planets:
            ang = ph + t * np.sqrt((m_star + mp) / a ** 3)
            x.append(a * np.cos(ang)), y.append(a * np.sin(ang)), m.append(mp)
            rsm.append(0.6 * a * np.cbrt(mp / (3.0 * m_star)))
            ix -= mp * x[-1] / a ** 3     # acceleration of the star by the planet, as the indirect term
            iy -= mp * y[-1] / a ** 3
        return (x, y, m, rsm), (ix, iy)

    outs = []
    for L in (product, oracle):
        dd = d.copy()
        dd.rank, dd.nranks = 0, 1
        radii = L.radii(dd)
        from tests.util import perturb
        fields = perturb(L.initial_fields(dd, radii), dd, 1e-3)
        (x, y, m, rsm), ind = bodies_at(0.0)
        ctx = driver.make_context(L, dd, fields=fields, radii=radii, bodies=(x, y, m))
        ctx.set_bodies(x, y, m, rsm, ind)
        S = driver.SlabSet([ctx])
        S.prepare()
        t, dts = 0.0, []
        for _ in range(14):
            dt = S.calculate_timestep()
            (x, y, m, rsm), ind = bodies_at(t)
            ctx.set_bodies(x, y, m, rsm, ind)
            if leapfrog:
                (xm, ym, mm, rm), _ = bodies_at(t + 0.5 * dt)
                ctx.set_bodies_midstep(xm, ym, mm, rm)
            ctx.step(dt)
            ctx.post(dt)
            t += dt
            dts.append(dt)
        outs.append((S.gather(), dts))
        ctx.close()
    _check(outs, ("sigma", "vrad", "vazi", "energy") if adiabatic else ("sigma", "vrad", "vazi"))
    # the potential really moved: the same run with the bodies frozen at t = 0 differs
    frozen = run_pair(product, product, d, 14, bodies=bodies_at(0.0)[0][:3], nslabs=(1, 0))[0][0]
    assert rel_err(outs[0][0]["vrad"], frozen["vrad"]) > 1e-6


@pytest.mark.parametrize("adiabatic", [False, True])
def test_recalculate_derived_and_device_pointers(product, oracle, adiabatic):
    """restart_load's tail (restart.cpp:113): new state grids uploaded, then recalculate_derived_disk_quantities
    (SourceEuler.cpp:225-249) -- P, T, c_s, H, nu against the oracle; and fcpt_device_ptr: the addresses handed to
    callers that keep their data on the device really are the grids fcpt_download reads."""
    import torch
    from fargocpt_amd import driver
    d = setups.planet_disk(product, 40, 288, adiabatic=adiabatic)
    d.rank, d.nranks = 0, 1
    radii = product.radii(d)
    fields = product.initial_fields(d.copy(), radii)
    rng = np.random.default_rng(3)
    new = [f * (1.0 + 0.2 * rng.random(f.shape)) for f in fields]
    derived = {}
    for name, L in (("hip", product), ("oracle", oracle)):
        ctx = driver.make_context(L, d, fields=fields, radii=radii)
        for fid, arr in zip((B.F_SIGMA, B.F_VRAD, B.F_VAZI, B.F_ENERGY), new):
            ctx.upload(fid, arr)
        ctx.recalculate_derived()
        derived[name] = {f: ctx.download(f) for f in (B.F_PRESSURE, B.F_TEMPERATURE, B.F_SOUNDSPEED, B.F_SCALE_HEIGHT, B.F_VISCOSITY)}
        if name == "hip":
            ptr, count = ctx.device_ptr(B.F_VRAD)
            assert count == new[1].size

            class _View:   # torch adopts foreign device memory through the CUDA array interface
                __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}
            onchip = torch.as_tensor(_View(), device="cuda")
            ctx.synchronize()
            assert np.array_equal(onchip.cpu().numpy().reshape(new[1].shape), new[1])
        ctx.close()
    for f, a in derived["hip"].items():
        assert rel_err(a, derived["oracle"][f]) <= 1e-13, f


@pytest.mark.parametrize("nr,nphi,adiabatic,leapfrog", [(64, 320, False, False), (48, 96, True, False), (40, 3, False, True),
                                                        (198, 1, False, False)])
def test_massflow_grid(product, oracle, nr, nphi, adiabatic, leapfrog):
    """WriteMassFlow: the MASSFLOW grid (VanLeerRadial's density flux accumulated step by step,
    TransportEuler.cpp:609-616) after 12 steps, on the marching-kernel path, the narrow-ring path and a
    one-cell ring."""
    from fargocpt_amd import driver
    d = setups.planet_disk(product, nr, nphi, adiabatic=adiabatic) if nphi > 1 else setups.steady_state_accretion(product, nr, nphi)
    d.write_massflow = 1
    if leapfrog:
        d.integrator = B.INTEGRATOR_LEAPFROG
    got = []
    for L in (product, oracle):
        ctx = driver.make_context(L, d, bodies=setups.jupiter_bodies(d) if nphi > 1 else None)
        S = driver.SlabSet([ctx])
        S.prepare()
        S.run(12)
        got.append((ctx.download(B.F_MASSFLOW), ctx.state()))
        ctx.upload(B.F_MASSFLOW, np.zeros((nr + 1, nphi)))   # clear_after_write
        assert not ctx.download(B.F_MASSFLOW).any()
        ctx.close()
    (ma, sa), (mb, sb) = got
    assert np.abs(mb).max() > 0 and not mb[0].any() and not mb[nr].any()
    assert rel_err(ma, mb) <= TOL
    for k in sa:
        assert rel_err(sa[k], sb[k]) <= TOL, k
    # without the switch the grid does not exist
    d.write_massflow = 0
    ctx = driver.make_context(product, d)
    with pytest.raises(B.FcptError, match="FCPT_EINVAL"):
        ctx.download(B.F_MASSFLOW)
    ctx.close()


@pytest.mark.parametrize("case", ["shocktube_256x4", "ring_128x384", "ideal_48x96", "leapfrog_64x320", "iso_96x512"])
def test_graph_replay_is_bit_identical(product, case):
    """fcpt_run_steps replaying a captured hipGraph of two steps (launch-bound grids) against the same loop with
    plain launches: same bits, same clock -- also across a change of the bodies between two calls, which makes the
    captured launch arguments stale (the graph must be dropped and captured again)."""
    from fargocpt_amd import driver
    bodies = None
    if case == "shocktube_256x4":
        d = setups.shocktube(product, 256, 4, "SN")
        d.first_dt = 1e-6
    elif case == "ring_128x384":
        d = setups.spreading_ring(product, 128, 384)
        d.first_dt = 1e-3
    elif case == "ideal_48x96":
        d = setups.planet_disk(product, 48, 96, adiabatic=True)
        bodies = setups.jupiter_bodies(d)
    elif case == "leapfrog_64x320":
        d = setups.planet_disk(product, 64, 320)
        d.integrator = B.INTEGRATOR_LEAPFROG
        bodies = setups.jupiter_bodies(d)
    else:
        d = setups.planet_disk(product, 96, 512)
        bodies = setups.jupiter_bodies(d)
    out = []
    for graph in (1, 0):
        ctx = driver.make_context(product, d, bodies=bodies)
        ctx.set_option("graph_steps", graph)
        S = driver.SlabSet([ctx])
        S.prepare()
        assert ctx.run_steps(21) == 21          # odd: one plain step behind the replays
        if bodies is not None:
            x, y, m = bodies
            ctx.set_bodies([x[0], 0.0], [y[0], 1.0], m)   # the planet a quarter orbit further
        assert ctx.run_steps(12) == 12
        assert ctx.run_steps(3) == 3            # too short for a graph: plain launches
        c = ctx.clock
        out.append((ctx.state(), (c.time, c.last_dt, c.n_hydro_iter)))
        ctx.close()
    assert out[0][1] == out[1][1]
    assert out[0][1][2] == 36
    for k in out[0][0]:
        assert np.array_equal(out[0][0][k], out[1][0][k]), k


@pytest.mark.parametrize("case", ["iso_96x512", "ideal_61x320_sn", "iso_reference_bc_40x263"])
def test_transport_chunk_lengths_never_change_a_result(product, case):
    """The fused transport kernel divides the slab into chunks of rings, each marched by its own wavefronts; the
    library grades their lengths (long ones first) on grids that need more than one round of the GPU's wavefront
    slots.  Every ring is computed by exactly one chunk from the same operands, so ANY list of lengths gives the
    bits of the equal chunks -- here ragged lists (single rings, a chunk longer than the slab's remainder, the last
    entry repeating), damping zones at both ends, against transport_graded = 0."""
    from fargocpt_amd import driver
    if case == "iso_96x512":
        d, lengths = setups.planet_disk(product, 96, 512), [7, 7, 7, 1, 5, 13, 3]
    elif case == "ideal_61x320_sn":
        d, lengths = setups.planet_disk(product, 61, 320, adiabatic=True), [30, 1, 1, 2, 9]
        d.artificial_viscosity = B.ARTVISC_SN
    else:
        d, lengths = setups.planet_disk(product, 40, 263), [4, 100]
        for s in (0, 1):
            d.bc_sigma[s] = d.bc_energy[s] = d.bc_vrad[s] = d.bc_vaz[s] = B.BC_REFERENCE
    bodies = setups.jupiter_bodies(d)
    out = []
    for explicit in (True, False):
        ctx = driver.make_context(product, d, bodies=bodies)
        if explicit:
            ctx.set_transport_chunks(lengths)
            tab = ctx.transport_chunks()
            live = tab[tab[:, 2] > tab[:, 1]]
            tiles = -(-d.nphi // 53)
            assert len(live) >= (len(lengths) - 1) * tiles
            cover = np.zeros((tiles, d.nr_global), dtype=np.int32)      # every ring of every tile exactly once
            for tl, a, b in live:
                cover[tl, a:b] += 1
            assert (cover == 1).all()
        else:
            ctx.set_option("transport_graded", 0)
            assert len(ctx.transport_chunks()) == 0
        S = driver.SlabSet([ctx])
        S.prepare()
        assert ctx.run_steps(15) == 15
        if explicit:
            ctx.set_transport_chunks([])           # back to the built-in choice (equal chunks on a grid this small)
            assert len(ctx.transport_chunks()) == 0
        assert ctx.run_steps(4) == 4
        c = ctx.clock
        out.append((ctx.state(), (c.time, c.last_dt, c.n_hydro_iter)))
        ctx.close()
    assert out[0][1] == out[1][1]
    for k in out[0][0]:
        assert np.array_equal(out[0][0][k], out[1][0][k]), k


def test_heating_grids_after_the_device_loop(product):
    """fcpt_run_steps lets only its LAST step write Q+ and Q- (outputs of the ideal-EOS source step; the steps before
    leave their difference for the CFL kernel and save two grids of traffic): after the call the grids must be those
    of the last step, bit for bit what the host-driven loop leaves -- also when the next call is one step long."""
    from fargocpt_amd import driver
    d = setups.planet_disk(product, 72, 320, adiabatic=True)
    bodies = setups.jupiter_bodies(d)
    out = []
    for device_loop in (True, False):
        ctx = driver.make_context(product, d, bodies=bodies)
        S = driver.SlabSet([ctx])
        S.prepare()
        if device_loop:
            assert ctx.run_steps(7) == 7
            mid = (ctx.download(B.F_QPLUS).copy(), ctx.download(B.F_QMINUS).copy())
            assert ctx.run_steps(1) == 1
        else:
            S.run(7)
            mid = (ctx.download(B.F_QPLUS).copy(), ctx.download(B.F_QMINUS).copy())
            S.run(1)
        out.append((mid, ctx.download(B.F_QPLUS).copy(), ctx.download(B.F_QMINUS).copy(), ctx.state()))
        ctx.close()
    for a, b in zip(out[0][0], out[1][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert np.abs(out[0][1]).max() > 0.0
    for k in out[0][3]:
        assert np.array_equal(out[0][3][k], out[1][3][k]), k


@pytest.mark.parametrize("nslabs", [1, 2])
def test_cfl_thermal_option(product, oracle, nslabs, monkeypatch):
    """Option cfl_thermal (off by default: measured slower at 2048 x 4096): the marching transport stores the
    cell-local part of the ideal-EOS CFL sum with the new state and k_cfl_rings reads it instead of Sigma, e, Q+, Q-.
    Same dt history and fields as the oracle, on one slab and on two."""
    monkeypatch.setenv("FCPT_CFL_THERMAL", "1")
    d = setups.planet_disk(product, 80, 320, adiabatic=True)
    _check(run_pair(product, oracle, d, 25, bodies=setups.jupiter_bodies(d), nslabs=(nslabs, 1)), ("sigma", "vrad", "vazi", "energy"))


@pytest.mark.parametrize("case", ["iso_march_64x320", "iso_loops_48x64", "ideal_march_48x288_cubic", "ideal_cooling_40x256",
                                  "leapfrog_40x192", "stab_two_slabs_64x256", "ideal_loops_32x96"])
def test_body_force_from_accelerations(product, oracle, case):
    """BodyForceFromPotential: no: CalculateAccelOnGas (Pframeforce.cpp:96-189) fills
    ACCEL_RADIAL / ACCEL_AZIMUTHAL and the source step takes -(a(i) + a(i-1)) / 2 and -(a(j) + a(j-1)) / 2 instead of
    the potential's gradient (SourceEuler.cpp:348-353, 406-411) -- in the marching kernels (a template flag), in the
    per-loop kernels of narrow rings, with the planet's cubic smoothing, the energy equation, leapfrog (mid-step
    bodies) and radial slabs.  The state after N steps and the acceleration grids themselves against the oracle."""
    from fargocpt_amd import driver
    adi = case.startswith("ideal")
    dims = [t for t in case.split("_") if "x" in t and t[0].isdigit()][0]
    nr, nphi = (int(x) for x in dims.split("x"))
    d = setups.planet_disk(product, nr, nphi, adiabatic=adi)
    d.body_force_from_potential = 0
    bodies = list(setups.jupiter_bodies(d))
    fields = ("sigma", "vrad", "vazi") + (("energy",) if adi else ())
    nslabs = (1, 1)
    if "cubic" in case:       # Klahr & Kley smoothing inside the Hill radius (the derivative form of it) + an indirect term
        x, y, m = bodies[0], bodies[1], bodies[2]
        bodies = [x, y, m, [0.0, (m[1] / 3.0) ** (1.0 / 3.0)], (1.0e-4, -2.0e-4)]
    if "cooling" in case:
        d.cooling_surface, d.opacity, d.kappa_const = 1, B.OPACITY_CONST, 1.0e4
    if "leapfrog" in case:
        d.integrator = B.INTEGRATOR_LEAPFROG
    if "stab" in case:
        d.stabilize_viscosity = 1
        nslabs = (2, 1)
    _check(run_pair(product, oracle, d, 20, bodies=tuple(bodies), nslabs=nslabs), fields)
    # the grids CalculateAccelOnGas leaves behind, cell by cell (rows 0 and Nr stay zero)
    radii = product.radii(d)
    got = []
    for lib in (product, oracle):
        ctx = driver.make_context(lib, d, radii=radii, bodies=tuple(bodies))
        S = driver.SlabSet([ctx])
        S.prepare()
        S.run(2)
        got.append((ctx.download(B.F_ACCEL_RADIAL), ctx.download(B.F_ACCEL_AZIMUTHAL)))
        ctx.close()
    for a, b in zip(*got):
        assert a.shape == (nr + 1, nphi)
        assert not a[0].any() and not a[nr].any() and np.abs(b[1:nr]).max() > 0
        assert rel_err(a, b) <= 1e-12
    # and the grids do not exist without the switch
    d.body_force_from_potential = 1
    ctx = driver.make_context(product, d, radii=radii, bodies=tuple(bodies))
    with pytest.raises(B.FcptError, match="FCPT_EINVAL"):
        ctx.download(B.F_ACCEL_RADIAL)
    ctx.close()


@pytest.mark.parametrize("law", ["lin", "bell"])
def test_tabulated_opacity_laws_over_all_regions(product, oracle, law):
    """Opacity: Lin | Bell (src/opacity.cpp:45-297) in cooling_terms(): Q- of the init call and the state after five
    steps of a disk whose cells sweep rho = 1e-13 .. 1e-4 g/cm^3, T = 10 .. 1e7 K (every region of both laws)."""
    from fargocpt_amd import driver
    from tests.opacity_cases import sweep_state
    d, radii, fields, _ = sweep_state(product, B.OPACITY_LIN if law == "lin" else B.OPACITY_BELL, nr=40, nphi=192)
    got = []
    for lib in (product, oracle):
        ctx = driver.make_context(lib, d, fields=fields, radii=radii)
        got.append(ctx.download(B.F_QMINUS))
        ctx.close()
    assert (got[1][1:-1] > 0).all()
    assert np.abs(got[0] / np.where(got[1] != 0, got[1], 1.0) - (got[1] != 0)).max() <= 1e-11


def test_half_limiter_edge_cases(product):
    """The branch-free van Leer half slope of the transport kernels (max(ab, 0) times a guarded reciprocal) against
    the select form `ab > 0 ? ab / (a + b) : 0` of TransportEuler.cpp:306-312 on the inputs where they could part:
    +-0, +-denormal sums, sums that cancel, huge and tiny operands of either sign (a negative
    denormal sum gave NaN).  Ordinary operands: <= 2 ulp (reciprocal + Newton step instead of the division)."""
    rng = np.random.default_rng(5)
    tiny = np.array([0.0, -0.0, 5e-324, -5e-324, 1e-310, -1e-310, 2.2250738585072014e-308, -2.2250738585072014e-308,
                     1e-300, -1e-300, 1e-200, -1e-200, 1.0, -1.0, 1e150, -1e150, 1e300, -1e300])
    a, b = (x.ravel() for x in np.meshgrid(tiny, tiny))
    ra = rng.standard_normal(4096) * 10.0 ** rng.uniform(-30, 30, 4096)
    rb = ra * rng.uniform(-2.0, 3.0, 4096)
    a, b = np.concatenate([a, ra, ra, 1.0 + rng.uniform(-1e-15, 1e-15, 64)]), np.concatenate([b, rb, -ra, -np.ones(64)])
    got = product.selftest_half_limiter(B.LIMITER_VANLEER, a, b)
    with np.errstate(all="ignore"):
        ab = a * b
        want = np.where(ab > 0, ab / (a + b), 0.0)
    fin = np.isfinite(want)                      # (ab overflows for |a|, |b| ~ 1e300 in either form)
    bad = fin & ~np.isfinite(got)
    assert not bad.any(), (a[bad], b[bad], got[bad])
    zero = want == 0.0
    assert (got[zero] == 0.0).all()
    ok = fin & (np.abs(want) > 1e-290)           # (results in the denormal range carry fewer bits in either form)
    assert np.abs(got[ok] / want[ok] - 1.0).max() <= 4e-15   # v_rcp_f64 + one Newton step + the product: a few ulp


@pytest.mark.parametrize("case", ["iso_96x512", "ideal_64x384", "iso_graph_40x256"])
def test_boundary_call_inside_the_cfl_launch(product, case):
    """fcpt_run_steps on large grids lets the final boundary call of a step ride in the next step's CFL launch
    (k_cfl_rings_bc: the boundary workgroups first, the four rings that read what they write last, behind a stamp).
    Forced here on small grids (option bc_in_cfl = 2), with and without hipGraph replay: the same bits as the
    host-driven loop cfl -> calculate_timestep -> step -> post."""
    from fargocpt_amd import driver
    dims = case.split("_")[-1]
    nr, nphi = (int(x) for x in dims.split("x"))
    d = setups.planet_disk(product, nr, nphi, adiabatic=case.startswith("ideal"))
    radii = product.radii(d)
    fields = product.initial_fields(d.copy(), radii)
    states = []
    for mode in ("host", "merged"):
        ctx = driver.make_context(product, d, fields=fields, radii=radii, bodies=setups.jupiter_bodies(d))
        ctx.set_option("graph_steps", 1 if "graph" in case else 0)
        S = driver.SlabSet([ctx])
        S.prepare()
        if mode == "host":
            S.run(23)
        else:
            ctx.set_option("bc_in_cfl", 2)
            names = product.kernel_names()
            ctx.profile_start([names.index("k_cfl_rings_bc"), names.index("k_boundary")], max_launches=64)
            ctx.run_steps(9)      # profiled: plain launches
            prof = ctx.profile_stop()
            # eight of nine boundary calls rode in a CFL launch, the last was flushed (+ the nine pre-transport calls
            # where the source march does not fold them in)
            # (k_boundary: the flush, + the first step's pre-transport call after the upload of the state)
            assert prof["k_cfl_rings_bc"][1] == 8 and prof["k_boundary"][1] in (1, 2, 10, 11), prof
            ctx.run_steps(14)     # (graph case: two lead steps, captured cycles, the flush)
            if "graph" in case:
                assert ctx.get_option("graph_replays") > 0
        states.append((ctx.state(), ctx.clock.time))
        ctx.close()
    assert states[0][1] == states[1][1]
    for k in states[0][0]:
        assert np.array_equal(states[0][0][k], states[1][0][k]), k
