"""Descriptor presets: the reference's YAML setups as fcpt_desc values.

Each preset mirrors a setup file of the reference (cited), with the grid size
left free.  All numbers are code units (G = M* = 1, L0 = 1 au).
"""
from __future__ import annotations

from . import binding as B

# code-unit conversion of the reference's default units (src/units.cpp:158-185)
_G_CGS, _KB, _MU = 6.67430e-8, 1.380649e-16, 1.66053906660e-24
_L0, _M0 = 1.495978707e13, 1.988409870698051e33
_T0 = (_L0 ** 3 / (_G_CGS * _M0)) ** 0.5              # seconds per code time
TEMP0_K = _G_CGS * _MU / _KB * _M0 / _L0            # Kelvin per code temperature
SIGMA_CGS = _M0 / (_L0 * _L0)                        # g/cm^2 per code surface density
M_JUP = 9.547919e-4                                  # jupiterMass / solMass


def _composite(d: B.Desc, side: int, name: str):
    """InnerBoundary/OuterBoundary composites (src/boundary_conditions/config.cpp:345-440)."""
    name = name.lower()
    d.bc_sigma[side] = B.BC_ZEROGRADIENT
    d.bc_energy[side] = B.BC_ZEROGRADIENT
    d.bc_vrad[side] = {"zerogradient": B.BC_ZEROGRADIENT, "outflow": B.BC_OUTFLOW,
                       "reflecting": B.BC_REFLECTING, "reference": B.BC_REFERENCE}[name]
    if name == "reference":
        d.bc_sigma[side] = d.bc_energy[side] = B.BC_REFERENCE
    d.bc_vaz[side] = B.BC_KEPLERIAN  # default of InnerBoundaryVazi (config.cpp:263,305)


def spreading_ring(lib: B.Library, nr=256, nphi=2) -> B.Desc:
    """test/spreading_ring/setup.yml (viscous stress + transport, isothermal, constant nu)."""
    d = lib.desc_default()
    d.nr_global, d.nphi = nr, nphi
    d.rmin, d.rmax, d.radial_spacing = 0.2, 1.8, B.SPACING_LOGARITHMIC
    d.ic = B.IC_SPREADING_RING
    d.sigma0 = 8.83829e+05 / SIGMA_CGS
    d.sigma_slope, d.sigma_floor = 0.0, 1.0e-8
    d.set_sigma0, d.disk_mass = 1, 1.0
    d.aspect_ratio, d.flaring_index = 0.0, 0.0
    d.constant_viscosity, d.viscous_alpha = 4.77e-5, 0.0
    d.artificial_viscosity = B.ARTVISC_NONE
    d.artificial_viscosity_dissipation = 0
    d.eos, d.adiabatic_index = B.EOS_ISOTHERMAL, 1.0
    d.heating_viscous = 0
    d.minimum_temperature, d.maximum_temperature = 1e-9 / TEMP0_K, 1e100 / TEMP0_K
    d.cfl, d.mu = 0.5, 1.0
    d.initialize_vradial_zero = 1
    d.thickness_smoothing = 0.0
    d.fast_transport = 1
    _composite(d, 0, "outflow")
    _composite(d, 1, "outflow")
    d.damping = 0
    d.omega_frame = 0.0
    d.nsnapshots, d.nmonitor, d.monitor_timestep = 1, 1, 314.159265359
    d.damping_time_radius_outer = d.rmax
    return d


def shocktube(lib: B.Library, nr=100, nphi=2, artvisc="SN", leapfrog=False) -> B.Desc:
    """test/shockTube/setups/shocktube_{SN,SN_LF,TW,TW_LF}.yml (the reference's TW setups are
    both leapfrog; shocktube_SN.yml is the Euler one)."""
    d = lib.desc_default()
    d.nr_global, d.nphi = nr, nphi
    d.rmin, d.rmax, d.radial_spacing = 1000.0, 1001.0, B.SPACING_ARITHMETIC
    d.ic = B.IC_SHOCKTUBE
    d.sigma0 = 8887231.453904748 / SIGMA_CGS
    d.sigma_slope, d.sigma_floor = 0.0, 1.0e-100
    d.aspect_ratio, d.flaring_index = 1.0, 0.5
    d.constant_viscosity, d.viscous_alpha = 0.0, 0.0
    d.artificial_viscosity = {"SN": B.ARTVISC_SN, "TW": B.ARTVISC_TW}[artvisc]
    d.artificial_viscosity_dissipation, d.artificial_viscosity_factor = 1, 1.41
    d.eos, d.adiabatic_index = B.EOS_IDEAL, 1.4
    d.heating_viscous = 1 if artvisc == "TW" else 0
    # init_shock_tube_test sets every unit factor to 1 (src/init.cpp:443-516): Kelvin = code
    d.minimum_temperature, d.maximum_temperature = 1e-9 / TEMP0_K, 1e100 / TEMP0_K
    d.cfl, d.mu = 0.5, 1.0
    d.thickness_smoothing = 0.6
    d.fast_transport = 1
    _composite(d, 0, "reflecting")
    _composite(d, 1, "reflecting")
    d.damping = 0
    d.omega_frame = 0.0
    d.nsnapshots, d.nmonitor, d.monitor_timestep = 1, 1, 0.228
    d.G = d.Rgas = 1.0
    d.damping_time_radius_outer = d.rmax
    d.integrator = B.INTEGRATOR_LEAPFROG if leapfrog else B.INTEGRATOR_EULER
    return d


def planet_disk(lib: B.Library, nr=128, nphi=384, adiabatic=False, damping=True,
                first_dt=1e-3) -> B.Desc:
    """examples/config.yml: locally isothermal (or ideal) disk, alpha = 1e-3, TW artificial
    viscosity, reflecting boundaries + wave damping, frame rotating with Omega = 1.
    FirstDT = 1e-3 skips the 1.1x ramp from the default 1e-9."""
    d = lib.desc_default()
    d.nr_global, d.nphi = nr, nphi
    d.rmin, d.rmax, d.radial_spacing = 0.4, 2.5, B.SPACING_LOGARITHMIC
    d.ic = B.IC_PROFILE
    d.sigma0 = 200.0 / SIGMA_CGS
    d.sigma_slope, d.sigma_floor = 0.5, 1e-9
    d.aspect_ratio, d.flaring_index = 0.05, 0.0
    d.viscous_alpha, d.constant_viscosity = 1.0e-3, 0.0
    d.artificial_viscosity = B.ARTVISC_TW
    d.artificial_viscosity_dissipation, d.artificial_viscosity_factor = 1, 1.41
    d.eos = B.EOS_IDEAL if adiabatic else B.EOS_ISOTHERMAL
    d.adiabatic_index, d.mu = 1.4, 2.35
    d.heating_viscous = 1
    d.minimum_temperature, d.maximum_temperature = 3.0 / TEMP0_K, 1e100 / TEMP0_K
    d.heating_cooling_cfl_limit = 1.0
    d.cfl, d.cfl_max_var, d.first_dt = 0.5, 1.1, first_dt
    d.thickness_smoothing = 0.6
    d.fast_transport = 1
    d.omega_frame = 1.0
    _composite(d, 0, "reflecting")
    _composite(d, 1, "reflecting")
    d.damping = 1 if damping else 0
    d.damping_inner_limit, d.damping_outer_limit = 1.10, 0.90
    d.damping_time_factor, d.damping_time_radius_outer = 1.0e-1, 2.5
    for arr in (d.damp_vrad, d.damp_vaz, d.damp_sigma, d.damp_energy):
        arr[0] = arr[1] = B.DAMP_REFERENCE
    d.nsnapshots, d.nmonitor, d.monitor_timestep = 50, 10, 0.628
    return d


def temperature_test(lib: B.Library, nr=100, nphi=2) -> B.Desc:
    """test/TemperatureTest/angelo.yml: viscously heated, radiatively cooled disk in thermal
    equilibrium (D'Angelo et al. 2003): ideal EOS, constant nu = 5e16 cm^2/s, kappa = 2e-6 T^2 cm^2/g
    (Opacity: Simple), SurfaceCooling: thermal, HeatingViscous, leapfrog, no artificial viscosity,
    100 x 2 cells on r in [1, 20] au."""
    d = lib.desc_default()
    d.nr_global, d.nphi = nr, nphi
    d.rmin, d.rmax, d.radial_spacing = 1.0, 20.0, B.SPACING_LOGARITHMIC
    d.ic = B.IC_PROFILE
    d.sigma0 = 197.0 / SIGMA_CGS
    d.sigma_slope, d.sigma_floor = 0.0, 1e-9
    d.mu = 2.35
    d.aspect_ratio = (352.0 / TEMP0_K * 1.0 / d.mu) ** 0.5   # Temperature0: 352 K (Interpret.cpp:194-197)
    d.flaring_index = 0.5
    d.viscous_alpha, d.constant_viscosity = 0.0, 5.0e16 / (_L0 * _L0 / _T0)
    d.artificial_viscosity, d.artificial_viscosity_dissipation = B.ARTVISC_NONE, 0
    d.artificial_viscosity_factor = 1.0
    d.eos, d.adiabatic_index = B.EOS_IDEAL, 1.4
    d.heating_viscous, d.heating_viscous_factor = 1, 1.0
    d.cooling_surface, d.cooling_radiative_factor = 1, 1.0
    d.opacity, d.kappa_const = B.OPACITY_SIMPLE, 17.770441374359926
    d.tau_factor, d.density_factor = 1.0, 2.0
    d.minimum_temperature, d.maximum_temperature = 3.0 / TEMP0_K, 1.0e7 / TEMP0_K
    d.cfl, d.heating_cooling_cfl_limit = 0.5, 1000.0
    d.thickness_smoothing = 0.0
    d.fast_transport = 1
    d.initialize_vradial_zero = 1
    _composite(d, 0, "reflecting")
    _composite(d, 1, "reflecting")
    d.damping = 1
    d.damping_inner_limit, d.damping_outer_limit = 1.10, 0.90
    d.damping_time_factor, d.damping_time_radius_outer = 3.0e-1, d.rmax
    for arr in (d.damp_vaz, d.damp_sigma, d.damp_energy):
        arr[0] = arr[1] = B.DAMP_NONE
    d.damp_vrad[0] = d.damp_vrad[1] = B.DAMP_ZERO
    d.omega_frame = 0.0
    d.integrator = B.INTEGRATOR_LEAPFROG
    d.nsnapshots, d.nmonitor, d.monitor_timestep = 10, 10, 6.28e2
    return d


def temperature_test_theory(d: B.Desc, radii_med, sigma_code):
    """Equilibrium temperature (code units) of test/TemperatureTest/check_results.py,
    T = sqrt(27/128 kappa0 nu / sigma_SB) Sigma Omega_K, evaluated with the given surface density."""
    kappa0 = d.kappa_const * d.temperature_cgs ** 2    # kappa = kappa0 T^2, T in code units
    omega = (d.G * d.hydro_center_mass / radii_med ** 3) ** 0.5
    # Q+ = 9/4 Sigma nu Omega^2 = Q- = 2 sigma T^4 / (3/8 tau), tau = tau_factor / density_factor kappa Sigma
    tau_fac = d.tau_factor / d.density_factor
    return (27.0 / 64.0 * tau_fac * kappa0 * d.constant_viscosity / d.sigma_sb) ** 0.5 * sigma_code * omega


def irradiation_test(lib: B.Library, nr=200, nphi=2):
    """test/irradiation/angelo.yml: a passive disk heated by the star (T = 10 000 K, R = 1 solRadius)
    and cooled through kappa = const (D'Angelo & Marzari 2012): returns (desc, bodies, irradiation)."""
    d = temperature_test(lib, nr, nphi)
    d.rmax = 100.0
    d.damping_time_radius_outer = d.rmax
    d.sigma0, d.sigma_slope, d.sigma_floor = 10.0 / SIGMA_CGS, 1.0, 1e-7
    d.flaring_index = 0.3
    d.constant_viscosity = 5.0e14 / (_L0 * _L0 / _T0)
    d.heating_viscous = 0
    d.opacity, d.kappa_const = B.OPACITY_CONST, 2.0e-6
    bodies = ([0.0], [0.0], [d.hydro_center_mass])
    irradiation = ([10000.0 / TEMP0_K], [6.957e10 / _L0], None)
    return d, bodies, irradiation


def jupiter_bodies(d: B.Desc):
    """Star + Jupiter of examples/config.yml for fcpt_set_bodies: the planet sits at
    (1, 0) in the frame rotating with OmegaFrame = 1 (no indirect term, no feedback)."""
    x = [0.0, 1.0]
    y = [0.0, 0.0]
    m = [d.hydro_center_mass, M_JUP]
    return x, y, m


def steady_state_accretion(lib: B.Library, nr=198, nphi=1) -> B.Desc:
    """test/steady_state_accretion/setup.yml: an alpha disk (alpha = 0.1, h = 0.005) with Sigma ~ r^-1/2 is a
    steady accretion flow of 3 pi Sigma nu = 1e-8 solMass/yr; outflow boundaries, Sigma and v_r damped to the
    initial profile near both edges, 198 x 1 cells on r in [10, 100] au, WriteMassFlow."""
    d = lib.desc_default()
    d.nr_global, d.nphi = nr, nphi
    d.rmin, d.rmax, d.radial_spacing = 10.0, 100.0, B.SPACING_LOGARITHMIC
    d.ic = B.IC_PROFILE
    d.sigma0 = 600.187 / SIGMA_CGS
    d.sigma_slope, d.sigma_floor = 0.5, 1e-8
    d.aspect_ratio, d.flaring_index = 0.005, 0.0
    d.viscous_alpha, d.constant_viscosity = 0.1, 0.0
    d.artificial_viscosity, d.artificial_viscosity_dissipation = B.ARTVISC_NONE, 1
    d.artificial_viscosity_factor = 1.41
    d.eos, d.adiabatic_index, d.mu = B.EOS_ISOTHERMAL, 1.4, 2.35
    d.heating_viscous = 0
    d.minimum_temperature, d.maximum_temperature = 3.0 / TEMP0_K, 1e100 / TEMP0_K
    d.cfl = 0.4
    d.thickness_smoothing = 0.0
    d.fast_transport = 1
    d.omega_frame = 0.0
    _composite(d, 0, "outflow")
    _composite(d, 1, "outflow")
    d.damping = 1
    d.damping_inner_limit, d.damping_outer_limit = 2.0, 0.64
    d.damping_time_factor, d.damping_time_radius_outer = 1.0e-2, d.rmax
    for arr in (d.damp_vaz, d.damp_energy):
        arr[0] = arr[1] = B.DAMP_NONE
    for arr in (d.damp_vrad, d.damp_sigma):
        arr[0] = arr[1] = B.DAMP_REFERENCE
    d.nsnapshots, d.nmonitor, d.monitor_timestep = 10, 1000, 31.41526e1
    d.write_massflow = 1
    return d


# 1e-8 solMass/yr in code units (code time = yr / 2 pi for l0 = 1 au, m0 = 1 solMass)
MDOT_STEADY_CODE = 1.0e-8 * (_T0 / 3.15576e7)


def cold_disk(lib: B.Library, planet=False):
    """test/cold_disk/setup.yml and test/cold_disk_planet/setup.yml: an inviscid power-law disk with the ideal EOS
    and neither heating nor cooling must keep its temperature profile (to 10 % over 20 orbits; with a 2e-5 planet,
    TW artificial viscosity and 100 orbits in the second setup).  Base length 30 au; the grid comes from
    `cps: 3` cells per scale height (Interpret.cpp:206-228).  Returns (desc, bodies) with bodies = [(a, mass,
    ramp-up time in orbits)] for driver.CircularOrbits."""
    import math
    d = lib.desc_default()
    temp0_k = TEMP0_K / 30.0   # l0 = 30 au
    d.rmin, d.rmax, d.radial_spacing = 0.4, 2.0, B.SPACING_LOGARITHMIC
    d.aspect_ratio, d.flaring_index = 0.05, 0.2857142857142857
    cps = 3.0
    d.nr_global = int(round(math.log(d.rmax / d.rmin) / math.log(1 + d.aspect_ratio / cps)))
    d.nphi = int(round(2 * math.pi / ((d.rmax / d.rmin) ** (1.0 / d.nr_global) - 1)))
    d.ic = B.IC_PROFILE
    d.sigma0, d.sigma_slope, d.sigma_floor = 0.005743125733951172, 1.0, 1e-7
    d.viscous_alpha, d.constant_viscosity = 0.0, 0.0
    if planet:
        d.artificial_viscosity, d.artificial_viscosity_dissipation, d.artificial_viscosity_factor = B.ARTVISC_TW, 1, 3.0
    else:
        d.artificial_viscosity, d.artificial_viscosity_dissipation, d.artificial_viscosity_factor = B.ARTVISC_NONE, 0, 1.41
    d.eos, d.adiabatic_index, d.mu = B.EOS_IDEAL, 1.4, 2.35
    d.heating_viscous = 0
    d.minimum_temperature, d.maximum_temperature = 3.0 / temp0_k, 1e100 / temp0_k
    d.cfl, d.cfl_max_var, d.first_dt, d.heating_cooling_cfl_limit = 0.5, 1.1, 1.0e-1, 1.0
    d.thickness_smoothing = 0.6
    d.fast_transport = 1
    d.omega_frame = 0.0
    _composite(d, 0, "reflecting")
    _composite(d, 1, "reflecting")
    d.damping = 1
    d.damping_inner_limit, d.damping_outer_limit = 1.311, 0.763
    d.damping_time_factor, d.damping_time_radius_outer = 0.05, d.rmax
    for arr in (d.damp_vrad, d.damp_vaz, d.damp_sigma, d.damp_energy):
        arr[0] = arr[1] = B.DAMP_REFERENCE
    d.monitor_timestep = 0.6283185307179586
    d.nmonitor, d.nsnapshots = (100, 10) if planet else (10, 20)
    bodies = [(0.0, d.hydro_center_mass, 0.0)] + ([(1.0, 2e-5, 10.0)] if planet else [])
    return d, bodies
