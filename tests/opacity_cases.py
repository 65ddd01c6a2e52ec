"""A disk state whose cells sweep the (density, temperature) plane of the tabulated opacity laws, and an
independent numpy statement of the two laws (written from the published fit formulas of Lin & Papaloizou 1985 and
Bell & Lin 1994 as the reference's src/opacity.cpp:45-297 tabulates them) for the double-entry check of the oracle."""
import numpy as np

from fargocpt_amd import binding as B, setups


def sweep_state(lib, opacity, nr=40, nphi=64):
    """Descriptor + (sigma, vrad, vazi, energy) whose cell (i, j) sits at rho = 10^(-13 .. -4) g/cm^3 (along phi) and
    T = 10^(1 .. 7) K (along r, staggered in phi): every region of both laws and their borders."""
    d = setups.planet_disk(lib, nr, nphi, adiabatic=True)
    d.cooling_surface, d.opacity = 1, opacity
    d.maximum_temperature = 1.0e9 / d.temperature_cgs
    radii = lib.radii(d)
    rmed = 2.0 / 3.0 * (radii[1:nr + 1] ** 3 - radii[:nr] ** 3) / (radii[1:nr + 1] ** 2 - radii[:nr] ** 2)
    i, j = np.meshgrid(np.arange(nr), np.arange(nphi), indexing="ij")
    rho_cgs = 10.0 ** (-13.0 + 9.0 * (j + 0.37 * (i % 3)) / nphi)
    t_cgs = 10.0 ** (1.0 + 6.0 * (i + (j % 7) / 7.0) / nr)
    T, rho = t_cgs / d.temperature_cgs, rho_cgs / d.density_cgs
    omega_k = np.sqrt(d.G * d.hydro_center_mass / rmed ** 3)[:, None]
    g = d.adiabatic_index
    H = np.sqrt(d.Rgas * T / d.mu) / omega_k          # c_s / (sqrt(gamma) Omega_K) with c_s^2 = gamma R T / mu
    sigma = rho * d.density_factor * H
    energy = sigma * T * d.Rgas / (d.mu * (g - 1.0))
    d.sigma0 = float(sigma.min())                      # the floors stay below every cell
    d.sigma_floor = 1e-9
    vrad = np.zeros((nr + 1, nphi))
    vazi = np.sqrt(d.G * d.hydro_center_mass / rmed)[:, None] * np.ones((1, nphi)) - d.omega_frame * rmed[:, None]
    return d, radii, (np.ascontiguousarray(sigma), vrad, np.ascontiguousarray(vazi), np.ascontiguousarray(energy)), rmed


def lin_numpy(rho, T):
    """Lin & Papaloizou (1985), cgs."""
    ts4 = 1e-4 * T
    d13 = np.cbrt(rho)
    d23 = d13 ** 2
    o5, o6, o7 = 2e4 * d23 * ts4 ** 3, 1e4 * d13 * ts4 ** 10, 1.5e10 * rho / ts4 ** 2.5
    hot567 = ((o6 ** 2 * o7 ** 2 / (o6 ** 2 + o7 ** 2)) ** 2 + (o5 / (1 + (ts4 / (1.1 * rho ** 0.04762)) ** 10)) ** 4) ** 0.25
    hot78 = (o7 ** 4 + 0.348 ** 4) ** 0.25
    o3h, o4h = 50.0 * ts4, 2e-2 * d23 / ts4 ** 9
    mid345 = (o4h ** 4 * o3h ** 4 / (o4h ** 4 + o3h ** 4) + (o5 / (1 + 6.561e-5 / ts4 ** 8)) ** 4) ** 0.25
    o1, o2, o3 = 2e-4 * T ** 2, 2e16 / T ** 7, 5e-3 * T
    cold123 = ((o1 ** 2 * o2 ** 2 / (o1 ** 2 + o2 ** 2)) ** 2 + (o3 / (1 + 1e22 / T ** 10)) ** 4) ** 0.25
    above234 = T > 1.6e3 * rho ** 4.44444444e-2
    above456 = T > 5.7e3 * rho ** 2.381e-2
    in567 = (T < 2.28e6 * rho ** 2.267e-1) | (rho <= 1e-10)
    return np.where(above234, np.where(above456, np.where(in567, hot567, hot78), mid345), cold123)


def bell_numpy(rho, T):
    """Bell & Lin (1994), cgs."""
    T = np.where(T < 1.0, 10.0, T)
    ts4 = 1e-4 * T
    d13 = np.cbrt(rho)
    d23 = d13 ** 2
    o5, o6, o7 = 1e4 * d23 * ts4 ** 3, 1e4 * d13 * ts4 ** 10, 1.5e10 * rho / ts4 ** 2.5
    hot567 = ((o6 ** 2 * o7 ** 2 / (o6 ** 2 + o7 ** 2)) ** 2 + (o5 / (1 + (ts4 / (1.1 * rho ** 0.04762)) ** 10)) ** 4) ** 0.25
    hot78 = (o7 ** 4 + 0.348 ** 4) ** 0.25
    o3h, o4h = 10.0 * np.sqrt(ts4), 2e-15 * rho / ts4 ** 24
    mid345 = (o4h ** 4 * o3h ** 4 / (o4h ** 4 + o3h ** 4) + (o5 / (1 + 6.561e-5 / ts4 ** 8 * 1e2 * d23)) ** 4) ** 0.25
    o1, o2, o3 = 2e-4 * T ** 2, 2e16 / T ** 7, 0.1 * np.sqrt(T)
    cold123 = ((o1 ** 2 * o2 ** 2 / (o1 ** 2 + o2 ** 2)) ** 2 + (o3 / (1 + 1e22 / T ** 10)) ** 4) ** 0.25
    above234 = T > 1.46e3 * rho ** 2.8369e-2
    above456 = T > 4.51e3 * rho ** 1.1464e-2
    in567 = (T < 2.37e6 * rho ** 2.2667e-1) | ((rho <= 1e10) & (T < 1e4))
    return np.where(above234, np.where(above456, np.where(in567, hot567, hot78), mid345), cold123)


def qminus_numpy(d, sigma, energy, rmed, law):
    """thermal_cooling / alpha at init (SourceEuler.cpp:790-820, 1507-1547) from the state, with the numpy law."""
    g = d.adiabatic_index
    T = d.mu * (g - 1.0) / d.Rgas * energy / sigma
    omega_k = np.sqrt(d.G * d.hydro_center_mass / rmed ** 3)[:, None]
    cs = np.sqrt(g * (g - 1.0) * energy / sigma)
    H = cs / np.sqrt(g) / omega_k
    rho = sigma / (d.density_factor * H)
    kappa = d.kappa_factor * law(rho * d.density_cgs, T * d.temperature_cgs) / d.opacity_cgs
    tau = d.tau_factor / d.density_factor * kappa * sigma
    tau_eff = 3.0 / 8.0 * tau + np.sqrt(3.0) / 4.0 + 1.0 / (4.0 * tau + d.tau_min)
    q = d.cooling_radiative_factor * 2 * d.sigma_sb * (T ** 4 - d.minimum_temperature ** 4) / tau_eff
    b = d.mu * (g - 1.0) / (d.Rgas * sigma)
    alpha = 1.0 + 2.0 * H * 4.0 * d.sigma_sb / d.c_light * b ** 4 * energy ** 3
    return q / alpha, kappa * d.opacity_cgs
