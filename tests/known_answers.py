"""The reference's own known-answer checks, restated (they are pure post-processing):
test/shockTube/check_results.py:93-127 and test/spreading_ring/calc_deviation.py:17-66."""
from __future__ import annotations

import os

import numpy as np
from scipy import integrate, interpolate
from scipy.special import iv

from fargocpt_amd import binding as B

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SHOCKTUBE_THRESHOLDS = {"vrad": 0.0153, "Sigma": 0.0073, "Temperature": 0.016, "energy": 0.014}
SPREADING_RING_THRESHOLD = 0.007


def shocktube_deviations(lib, d, ctx):
    an = np.loadtxt(os.path.join(GOLDEN, "shocktube_analytic_shock.dat"), skiprows=2)
    r12 = lib.radii(d)[:d.nr_global + 1]
    r1 = 0.5 * (r12[1:] + r12[:-1]) - r12[0]
    st = ctx.state()
    v = st["vrad"].mean(1)
    q = {"vrad": 0.5 * (v[1:] + v[:-1]), "Sigma": st["sigma"].mean(1),
         "Temperature": ctx.download(B.F_TEMPERATURE).mean(1), "energy": st["energy"].mean(1)}
    key = {"vrad": 0, "Sigma": 1, "Temperature": 2, "energy": 3}
    inds = (r1 >= 0) & (r1 <= 1)
    out = {}
    for name, data in q.items():
        y = an[:, key[name] + 2]
        if name == "energy":
            y = an[:, 4] * an[:, 3] / (1.4 - 1)
        spl = interpolate.InterpolatedUnivariateSpline(an[:, 1], y)
        out[name] = float(integrate.simpson(np.abs(data[inds] - spl(r1[inds])), x=r1[inds]))
    return out


def spreading_ring_deviation(lib, d, ctx):
    radii = lib.radii(d)[:d.nr_global + 1]
    Rinf, Rsup = radii[:-1], radii[1:]
    rc = 2.0 / 3.0 * (Rsup ** 3 - Rinf ** 3) / (Rsup ** 2 - Rinf ** 2)
    sigma = ctx.download(B.F_SIGMA).mean(1)
    t = ctx.clock.time
    nu, tau0 = 4.77e-5, 0.016
    tau = 12 * nu * t + tau0
    theo = 1.0 / np.pi / tau / rc ** 0.25 * iv(0.25, 2.0 * rc / tau) * np.exp(-(1 + rc ** 2) / tau)
    return float(np.mean(np.abs(sigma / theo - 1)))


STEADY_ACCRETION_THRESHOLD = 2.2e-4   # test/steady_state_accretion/testconfig.yml:2


def steady_accretion_deviation(lib, d, massflow_1d):
    """test/steady_state_accretion/check_results.py:104-118: max |MassFlow / 1e-8 solMass/yr| - 1 over the interfaces
    between 20 and 60 au.  `massflow_1d`: the Nr + 1 values of MassFlow1D.dat of the last snapshot (code units:
    azimuthal sum of MASSFLOW divided by Nmonitor * MonitorTimestep)."""
    from fargocpt_amd import setups
    radii = lib.radii(d)
    ri, rs = radii[:d.nr_global], radii[1:d.nr_global + 1]
    x = 2.0 / 3.0 * (rs ** 3 - ri ** 3) / (rs ** 2 - ri ** 2)   # Rmed: the radii of Sigma1D.dat
    diffval = np.abs(np.asarray(massflow_1d)[1:-1]) / setups.MDOT_STEADY_CODE - 1
    inds = np.logical_and(x[1:] > 20.0, x[:-1] < 60.0)
    return float(np.max(np.abs(diffval[inds])))


def run_steady_accretion(lib_run, lib_host, d):
    """The reference's run of that setup through the ABI: Nsnapshots x Nmonitor monitor steps with snapping; at
    every snapshot MASSFLOW is divided by Nmonitor * MonitorTimestep, summed over azimuth (MassFlow1D.dat) and
    cleared (quantities.cpp:770-781, data.cpp:276-278).  Returns (MassFlow1D of the last snapshot, hydro steps)."""
    from fargocpt_amd import driver
    ctx = driver.make_context(lib_run, d)
    S = driver.SlabSet([ctx])
    S.prepare()
    steps, mf = 0, None
    for snap in range(1, d.nsnapshots + 1):
        t_end = snap * d.nmonitor * d.monitor_timestep
        while ctx.clock.time < t_end * (1.0 - 1e-14):
            steps += ctx.run_steps(1, snap=True)
        mf = ctx.download(B.F_MASSFLOW).sum(axis=1) / (d.nmonitor * d.monitor_timestep)
        ctx.upload(B.F_MASSFLOW, np.zeros((d.nr_global + 1, d.nphi)))
    ctx.close()
    return mf, steps


COLD_DISK_THRESHOLD = 0.1   # test/cold_disk/calc_deviation.py:31-34 (and cold_disk_planet)


def run_cold_disk(lib_run, lib_host, planet, max_snapshots=None):
    """test/cold_disk(_planet)/setup.yml through the ABI with the driver's circular-orbit stand-in for the N-body
    system.  Returns a dict: hydro steps, the criterion of calc_deviation.py:22-34 -- max |<T>_phi(last) /
    <T>_phi(first) - 1| -- after every snapshot, and how non-axisymmetric the final density is."""
    from fargocpt_amd import driver, setups
    d, bodies = setups.cold_disk(lib_host, planet)
    orb = driver.CircularOrbits(d, bodies)
    ctx = driver.make_context(lib_run, d, bodies=orb.at(0.0, 0.0))
    T0 = ctx.download(B.F_TEMPERATURE).mean(axis=1)
    devs = []
    steps = driver.run_to_snapshots(
        ctx, d, orb, lambda n: devs.append(float(np.max(np.abs(ctx.download(B.F_TEMPERATURE).mean(axis=1) / T0 - 1)))),
        max_snapshots=max_snapshots)
    sig = ctx.download(B.F_SIGMA)
    out = {"grid": [d.nr_global, d.nphi], "steps": steps, "deviation_per_snapshot": devs,
           "sigma_nonaxisymmetry": float(np.max(np.abs(sig / sig.mean(axis=1, keepdims=True) - 1)))}
    ctx.close()
    return out
