"""Shared helpers of the parity tests."""
from __future__ import annotations

import numpy as np

from fargocpt_amd import binding as B, driver


def rel_err(a: np.ndarray, b: np.ndarray) -> float:
    """max|a-b| / max|b|: v_r has zeros, a pointwise relative error is meaningless
    (SURVEY.md section 8(c))."""
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0))


def perturb(fields, d, amp=1e-3):
    """Deterministic non-axisymmetric perturbation 1 + amp sin(3 phi) cos(5 ln r) of Sigma
    (and energy), so limiters, shifts and azimuthal fluxes are exercised
    (SURVEY.md section 8(d))."""
    sigma, vrad, vazi, energy = [f.copy() for f in fields]
    nr, nphi = sigma.shape
    phi = (np.arange(nphi) + 0.5) * 2 * np.pi / nphi
    r = np.geomspace(max(d.rmin, 1e-3), d.rmax, nr)
    f = 1.0 + amp * np.sin(3 * phi)[None, :] * np.cos(5 * np.log(r))[:, None]
    sigma *= f
    energy *= f
    vazi = vazi * (1.0 + 0.1 * amp * np.cos(2 * phi)[None, :])
    return sigma, vrad, vazi, energy


def run_pair(lib_a, lib_b, d, nsteps, bodies=None, amp=1e-3, snap=False, nslabs=(1, 1), dt_scale=1.0, noise=0.0,
             irradiation=None, transport_chunks=None):
    """Advance the same initial state `nsteps` with two libraries; returns the two global
    states and the two dt histories."""
    outs = []
    dfull = d.copy()
    dfull.rank, dfull.nranks = 0, 1
    radii = lib_a.radii(dfull)
    d0 = dfull.copy()
    fields = lib_a.initial_fields(d0, radii)   # d0.sigma0 possibly rescaled
    if amp:
        fields = perturb(fields, d0, amp)
    if noise:   # cell-wise relative noise (seeded): how fast does this flow amplify rounding-sized differences?
        rng = np.random.default_rng(7)
        fields = tuple(f * (1.0 + noise * rng.standard_normal(f.shape)) for f in fields)
    for L, ns in zip((lib_a, lib_b), nslabs):
        if ns == 0:   # only one library wanted
            outs.append(None)
            continue
        ctxs = []
        for rank in range(ns):
            dd = d0.copy()
            dd.rank, dd.nranks = rank, ns
            s = L.split_domain(dd)
            sl = slice(s.imin, s.imin + s.nr)
            sub = (fields[0][sl], fields[1][s.imin:s.imin + s.nr + 1], fields[2][sl], fields[3][sl])
            sub = tuple(np.ascontiguousarray(x) for x in sub)
            ctxs.append(driver.make_context(L, dd, fields=sub, radii=radii, bodies=bodies, irradiation=irradiation))
            if transport_chunks is not None and L.has("set_transport_chunks"):   # (the product's marching kernel; the oracle has no chunks)
                ctxs[-1].set_transport_chunks(transport_chunks)
        S = driver.SlabSet(ctxs)
        S.dt_scale = dt_scale
        S.prepare()
        dts = S.run(nsteps, snap=snap)
        outs.append((S.gather(), dts))
        for c in ctxs:
            c.close()
    return outs
