"""End-to-end drop-in test of the host driver `fargocpt_hip` (YAML setup in, reference-format
snapshot files out): the reference's own setup files are run on the GPU and the written files are
checked the way the reference's checkers do (test/shockTube/check_results.py:93-127,
test/spreading_ring/calc_deviation.py:8-66), with the reference's thresholds."""
import os
import struct
import subprocess

import numpy as np
import pytest
from scipy import integrate, interpolate
from scipy.special import iv

from tests.known_answers import GOLDEN, SHOCKTUBE_THRESHOLDS, SPREADING_RING_THRESHOLD

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "fargocpt_amd", "bin", "fargocpt_hip")


def _run(tmp_path, setup, outname):
    cfg = tmp_path / "config.yml"
    out = tmp_path / outname
    text = open(os.path.join(GOLDEN, "setups", setup)).read().splitlines()
    text = [("OutputDir: " + str(out)) if l.startswith("OutputDir") else l for l in text]
    cfg.write_text("\n".join(text) + "\n")
    r = subprocess.run([BIN, "-q", "start", str(cfg)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return str(out) + "/"


def _misc(path):
    raw = open(path, "rb").read()
    assert len(raw) == 48  # src/output.h:16-24
    snap, mon, time, omega, angle, last_dt, n_iter = struct.unpack("<IIdddd Q".replace(" ", ""), raw)
    return dict(snapshot=snap, monitor=mon, time=time, last_dt=last_dt, n_iter=n_iter)


@pytest.mark.parametrize("setup,steps", [("shocktube_SN.yml", 270), ("shocktube_SN_LF.yml", 235),
                                         ("shocktube_TW.yml", 243), ("shocktube_TW_LF.yml", 243)])
def test_shocktube_setup_files(tmp_path, setup, steps):
    out = _run(tmp_path, setup, "out")
    assert open(out + "snapshots/list.txt").read().split() == ["0", "1"]
    misc = _misc(out + "snapshots/1/misc.bin")
    assert misc["snapshot"] == 1 and misc["n_iter"] == steps and abs(misc["time"] - 0.228) < 1e-12
    # units.yml / info2D.yml as the reference's Python loader reads them (python_module/fargocpt/data.py:55-61,
    # 680-700): every grid is found through its info2D entry
    import yaml
    units = yaml.safe_load(open(out + "units.yml"))
    assert units["length"]["unit"].split()[1] == "cm" and abs(float(units["length"]["unit"].split()[0]) / 1.495978707e13 - 1) < 1e-12
    info = yaml.safe_load(open(out + "info2D.yml"))
    assert {"Sigma", "vrad", "vazi", "energy", "Temperature"} <= set(info)
    for name, meta in info.items():
        data = np.fromfile(out + "snapshots/1/" + meta["filename"])
        assert data.size == meta["Nrad"] * meta["Nazi"], name
        assert meta["on_radial_interface"] == (name == "vrad")
    # test/shockTube/check_results.py:93-127
    an = np.loadtxt(os.path.join(GOLDEN, "shocktube_analytic_shock.dat"), skiprows=2)
    r12 = np.loadtxt(out + "used_rad.dat")
    r1 = 0.5 * (r12[1:] + r12[:-1]) - r12[0]
    nr = len(r1)
    key = {"vrad": 0, "Sigma": 1, "Temperature": 2, "energy": 3}
    inds = (r1 >= 0) & (r1 <= 1)
    for quant, thr in SHOCKTUBE_THRESHOLDS.items():
        data = np.fromfile(out + f"snapshots/1/{quant}.dat")
        if quant == "vrad":
            data = data.reshape((nr + 1, -1)).mean(1)
            data = 0.5 * (data[1:] + data[:-1])
        else:
            data = data.reshape((nr, -1)).mean(1)
        y = an[:, key[quant] + 2]
        if quant == "energy":
            y = an[:, 4] * an[:, 3] / (1.4 - 1)
        spl = interpolate.InterpolatedUnivariateSpline(an[:, 1], y)
        diff = integrate.simpson(np.abs(data[inds] - spl(r1[inds])), x=r1[inds])
        assert diff < thr, (quant, diff, thr)


def test_spreading_ring_setup_file(tmp_path):
    out = _run(tmp_path, "spreading_ring.yml", "ring")
    nr, naz = np.genfromtxt(out + "dimensions.dat", usecols=(4, 5), unpack=True, dtype=int)
    ri = np.genfromtxt(out + "used_rad.dat")
    rc = 2.0 / 3.0 * (ri[1:] ** 3 - ri[:-1] ** 3) / (ri[1:] ** 2 - ri[:-1] ** 2)
    n = int(open(out + "snapshots/list.txt").read().split()[-1])
    misc = _misc(out + f"snapshots/{n}/misc.bin")
    assert misc["n_iter"] == 39870  # the reference's own step count for this setup
    sigma = np.fromfile(out + f"snapshots/{n}/Sigma.dat").reshape(nr, naz).mean(1)
    tau = 12 * 4.77e-5 * misc["time"] + 0.016
    theo = 1.0 / np.pi / tau / rc ** 0.25 * iv(0.25, 2.0 * rc / tau) * np.exp(-(1 + rc ** 2) / tau)
    assert np.mean(np.abs(sigma / theo - 1)) < SPREADING_RING_THRESHOLD


def test_temperature_test_setup_file(tmp_path):
    """test/TemperatureTest/angelo.yml through the driver on the GPU (577 k leapfrog steps of a 100 x 2
    grid, ~2 minutes: launch-bound), checked as test/TemperatureTest/check_results.py checks it, with
    that script's constants and threshold; the 1-D files it reads are azimuthal means of the 2-D ones."""
    cfgtext = open(os.path.join(GOLDEN, "setups", "temperature_test_angelo.yml")).read()
    cfg = tmp_path / "config.yml"
    out = tmp_path / "tt"
    lines = [("OutputDir: " + str(out)) if l.startswith("OutputDir") else l for l in cfgtext.splitlines()]
    cfg.write_text("\n".join(lines) + "\n")
    r = subprocess.run([BIN, "-q", "start", str(cfg)], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr
    out = str(out) + "/"
    assert open(out + "snapshots/list.txt").read().split()[-1] == "10"
    ri = np.genfromtxt(out + "used_rad.dat")
    r = 2.0 / 3.0 * (ri[1:] ** 3 - ri[:-1] ** 3) / (ri[1:] ** 2 - ri[:-1] ** 2)
    nr = len(r)
    quant1 = np.fromfile(out + "snapshots/10/Temperature.dat").reshape(nr, -1).mean(1)
    # --- check_results.py ---
    dens = 300 * np.sqrt(5 / r)
    kappa = 2e-6
    nu = 5e16
    sigma = 5.6704e-05
    l0 = 14959787070000
    m0 = 1.98892e+33
    Sigma0 = m0 / l0 / l0
    T0 = 1.0756431684186062e+05
    G = 6.674e-8
    omega_k = np.sqrt(G * m0 * (r * l0) ** (-3))
    Ttheo = np.sqrt(27 / 128 * kappa * nu / sigma) * dens * omega_k
    Tnum = quant1 * T0
    Tdiff = np.abs(Tnum - Ttheo) / Ttheo
    radial_range = np.logical_and(r > 2, r < 15)
    assert np.max(Tdiff[radial_range]) < 0.01
    densnum = np.fromfile(out + "snapshots/10/Sigma.dat").reshape(nr, -1).mean(1) * Sigma0
    assert np.max((np.abs(densnum - dens) / dens)[radial_range]) < 0.01


def test_irradiation_setup_file(tmp_path):
    """test/irradiation/angelo.yml through the driver on the GPU (168 k leapfrog steps of a 200 x 2 grid),
    checked with test/irradiation/check_results.py's constants and threshold."""
    from tests.test_oracle_known_answers import _irradiation_deviation
    cfgtext = open(os.path.join(GOLDEN, "setups", "irradiation_angelo.yml")).read()
    cfg = tmp_path / "config.yml"
    out = tmp_path / "irr"
    lines = [("OutputDir: " + str(out)) if l.startswith("OutputDir") else l for l in cfgtext.splitlines()]
    cfg.write_text("\n".join(lines) + "\n")
    r = subprocess.run([BIN, "-q", "start", str(cfg)], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr
    out = str(out) + "/"
    ri = np.genfromtxt(out + "used_rad.dat")
    rc = 2.0 / 3.0 * (ri[1:] ** 3 - ri[:-1] ** 3) / (ri[1:] ** 2 - ri[:-1] ** 2)
    T = np.fromfile(out + "snapshots/10/Temperature.dat").reshape(len(rc), -1).mean(1)
    assert _irradiation_deviation(rc, T) < 0.03
