"""Pins the CPU oracle with the reference's OWN known-answer tests (SURVEY.md section 8(c) ii):

* test/shockTube  : analytic Sod solution `analytic_shock.dat` + thresholds of
  check_results.py:18-23, Euler setups shocktube_SN.yml / shocktube_TW.yml;
* test/spreading_ring : analytic viscous-ring solution + threshold of calc_deviation.py:43-66.

Also pins the number of hydro steps the run takes: 270 for shocktube_SN and 39 870 for the
256x2 spreading ring are the counts the compiled reference produced,
so the oracle reproduces the reference's dt history, not only its end state."""
import json
import os

import numpy as np
import pytest

from fargocpt_amd import binding as B, driver, setups
from tests.known_answers import (GOLDEN, SHOCKTUBE_THRESHOLDS, SPREADING_RING_THRESHOLD,
                                 shocktube_deviations, spreading_ring_deviation)

GOLD = json.load(open(os.path.join(GOLDEN, "oracle_reference_runs.json")))


def _run_to_snapshot(lib, d):
    ctx = driver.make_context(lib, d)
    s = driver.SlabSet([ctx])
    s.prepare()
    n = ctx.run_steps(200000, snap=True)
    return ctx, n


@pytest.mark.parametrize("av,lf", [("SN", False), ("TW", False), ("SN", True), ("TW", True)])
def test_shocktube_analytic(product, oracle, av, lf):
    """SN Euler = shocktube_SN.yml, SN/TW leapfrog = shocktube_SN_LF.yml / shocktube_TW(_LF).yml
    (step_LeapFrog, src/simulation.cpp:276-459); TW Euler is an extra combination."""
    d = setups.shocktube(product, 100, 2, av, leapfrog=lf)
    ctx, n = _run_to_snapshot(oracle, d)
    key = f"shocktube_{av}" + ("_LF" if lf else "")
    assert n == GOLD[key]["steps"]
    if (av, lf) == ("SN", False):
        assert n == 270  # the reference's own step count
    assert abs(ctx.clock.time - 0.228) < 1e-12
    dev = shocktube_deviations(product, d, ctx)
    for k, thr in SHOCKTUBE_THRESHOLDS.items():
        assert dev[k] < thr, (k, dev[k], thr)
        assert dev[k] == pytest.approx(GOLD[key]["deviations"][k], rel=1e-9)
    st = ctx.state()
    assert float(st["sigma"].sum()) == pytest.approx(GOLD[key]["sum_sigma"], rel=1e-12)


def test_spreading_ring_analytic(product, oracle):
    d = setups.spreading_ring(product, 256, 2)
    ctx, n = _run_to_snapshot(oracle, d)
    assert n == 39870  # the reference's own step count for this setup
    dev = spreading_ring_deviation(product, d, ctx)
    assert dev < SPREADING_RING_THRESHOLD
    assert dev == pytest.approx(GOLD["spreading_ring_256x2"]["mean_rel_deviation"], rel=1e-9)


def test_mass_conservation_closed_box(product, oracle):
    """Reflecting boundaries, no damping: the transport is conservative, total mass of the
    active rings changes only through the ghost rings (secondary parity metric)."""
    d = setups.planet_disk(product, 48, 96, damping=False)
    ctx = driver.make_context(oracle, d)
    radii = product.radii(d)[:d.nr_global + 1]
    surf = np.pi * (radii[1:] ** 2 - radii[:-1] ** 2) / d.nphi
    m0 = (ctx.download(B.F_SIGMA)[1:-1] * surf[1:-1, None]).sum()
    s = driver.SlabSet([ctx])
    s.prepare()
    s.run(50)
    m1 = (ctx.download(B.F_SIGMA)[1:-1] * surf[1:-1, None]).sum()
    assert abs(m1 / m0 - 1) < 1e-11


def test_one_vs_many_slabs_oracle(product, oracle):
    """Radial decomposition: the reference agrees to 4e-13 between 1 and 2 ranks; the oracle's
    slabs (7-ring overlap, commbound exchange) reproduce the single-slab run bit for bit."""
    from tests.util import rel_err, run_pair
    d = setups.planet_disk(product, 96, 64, adiabatic=True)
    (a, dta), (b, dtb) = run_pair(oracle, oracle, d, 25, nslabs=(3, 1), bodies=setups.jupiter_bodies(d))
    assert dta == dtb
    for k in a:
        assert rel_err(a[k], b[k]) == 0.0


def test_disk_on_body_accel_symmetries(product, oracle):
    """ComputeDiskOnPlanetAccel restated (Force.cpp:23-122): for an axisymmetric disk the force on an
    object on the x axis has no y component, the rings inside pull inward and the rings outside pull
    outward; the force on the star vanishes."""
    from fargocpt_amd import driver, setups
    d = setups.planet_disk(product, 40, 96)
    ctx = driver.make_context(oracle, d, bodies=setups.jupiter_bodies(d))
    axi, ayi, axo, ayo = ctx.disk_on_body_accel(1.0, 0.0, 1.0)
    assert axi < 0.0 < axo
    assert abs(ayi) < 1e-12 * abs(axi) and abs(ayo) < 1e-12 * abs(axo)
    star = ctx.disk_on_body_accel(0.0, 0.0, 0.0, 0.0, 0.0)
    assert np.all(np.abs(star) < 1e-12 * abs(axo))
    # planet-location smoothing weakens the pull of the nearby rings
    sm = ctx.disk_on_body_accel(1.0, 0.0, 1.0, 0.3)
    assert abs(sm[0]) < abs(axi) and abs(sm[2]) < abs(axo)
    ctx.close()
