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

from tests.known_answers import GOLDEN, SHOCKTUBE_THRESHOLDS, SPREADING_RING_THRESHOLD, STEADY_ACCRETION_THRESHOLD

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


def test_steady_state_accretion_setup_file(tmp_path, product):
    """test/steady_state_accretion/setup.yml (198 x 1 cells, WriteMassFlow) through the driver: MassFlow1D.dat of the
    last snapshot against the reference's criterion (check_results.py:104-118, threshold 2.2e-4) and against the
    oracle's run of the same setup (tests/golden/oracle_reference_runs.json: same step count, same deviation)."""
    import json
    from fargocpt_amd import setups
    from tests.known_answers import steady_accretion_deviation
    out = _run(tmp_path, "steady_state_accretion.yml", "acc")
    gold = json.load(open(os.path.join(GOLDEN, "oracle_reference_runs.json")))["steady_state_accretion_198x1"]
    snaps = open(out + "snapshots/list.txt").read().split()
    assert snaps == [str(n) for n in range(11)]
    misc = _misc(out + "snapshots/10/misc.bin")
    assert misc["n_iter"] == gold["steps"]
    pairs = np.fromfile(out + "snapshots/10/MassFlow1D.dat").reshape(-1, 2)
    assert pairs.shape[0] == 199
    assert np.allclose(pairs[:, 0], np.loadtxt(out + "used_rad.dat"), rtol=1e-15)
    d = setups.steady_state_accretion(product)
    dev = steady_accretion_deviation(product, d, pairs[:, 1])
    assert dev < STEADY_ACCRETION_THRESHOLD
    assert dev == pytest.approx(gold["max_rel_deviation"], rel=1e-6)
    assert pairs[100, 1] == pytest.approx(gold["massflow_code_units_at_interface_100"], rel=1e-9)


@pytest.mark.parametrize("setup,key,nsnap", [("cold_disk.yml", "cold_disk", 20), ("cold_disk_planet.yml", "cold_disk_planet", 10)])
def test_cold_disk_setup_files(tmp_path, setup, key, nsnap):
    """test/cold_disk/setup.yml and test/cold_disk_planet/setup.yml (cps grid key, l0 = 30 au, planet with mass
    ramp-up on its circular orbit + indirect term) through the driver, checked as calc_deviation.py:22-60 does:
    dimensions.dat, snapshots/list.txt, units.yml and the Temperature.dat files; threshold 0.1.  The deviation and the
    step count equal the oracle's run of the same setup."""
    import json
    import yaml
    out = _run(tmp_path, setup, "cold")
    gold = json.load(open(os.path.join(GOLDEN, "oracle_reference_runs.json")))[key]
    Nr, Naz = np.genfromtxt(out + "dimensions.dat", usecols=(4, 5), unpack=True, dtype=int)
    assert [int(Nr), int(Naz)] == gold["grid"]
    Ns = np.genfromtxt(out + "snapshots/list.txt", dtype=int)
    assert list(Ns) == list(range(nsnap + 1))
    tempunit = yaml.safe_load(open(out + "units.yml"))["temperature"]["cgs value"]
    from fargocpt_amd import setups
    assert abs(tempunit / (setups.TEMP0_K / 30.0) - 1) < 1e-12   # l0 = 30 au
    prof = {n: (tempunit * np.fromfile(out + f"snapshots/{n}/Temperature.dat").reshape(Nr, Naz)).mean(axis=1) for n in (Ns[0], Ns[-1])}
    dev = np.max(np.abs(prof[Ns[-1]] / prof[Ns[0]] - 1))
    assert dev < 0.1
    assert dev == pytest.approx(gold["deviation_per_snapshot"][-1], rel=1e-4)
    assert _misc(out + f"snapshots/{nsnap}/misc.bin")["n_iter"] == gold["steps"]
    sig = np.fromfile(out + f"snapshots/{nsnap}/Sigma.dat").reshape(Nr, Naz)
    nonaxi = np.max(np.abs(sig / sig.mean(axis=1, keepdims=True) - 1))
    if key == "cold_disk_planet":
        assert nonaxi == pytest.approx(gold["sigma_nonaxisymmetry"], rel=1e-3)
    else:
        assert nonaxi < 1e-9


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


DISK_YML = """\
# a small planet disk for the restart test (own text; keys of the reference's config surface)
Nrad: 48
Naz: 320
Rmin: 0.4
Rmax: 2.5
RadialSpacing: Logarithmic
Sigma0: 200 g/cm2
SigmaSlope: 0.5
SigmaFloor: 1e-9
AspectRatio: 0.05
FlaringIndex: 0.0
ViscousAlpha: 1.0e-3
ArtificialViscosity: TW
ArtificialViscosityFactor: 1.41
ArtificialViscosityDissipation: Yes
EquationOfState: {eos}
AdiabaticIndex: 1.4
mu: 2.35
HeatingViscous: Yes
SurfaceCooling: {cooling}
Opacity: Lin
MinimumTemperature: 3 K
MaximumTemperature: 1e100 K
Integrator: {integrator}
Transport: FARGO
CFL: 0.5
FirstDT: 1e-3
OmegaFrame: 1.0
ThicknessSmoothing: 0.6
InnerBoundary: Reflecting
OuterBoundary: Reflecting
Damping: Yes
DampingInnerLimit: 1.1
DampingOuterLimit: 0.9
DampingTimeFactor: 0.1
DampingVRadialInner: Reference
DampingVRadialOuter: Reference
DampingVAzimuthalInner: Reference
DampingVAzimuthalOuter: Reference
DampingSurfaceDensityInner: Reference
DampingSurfaceDensityOuter: Reference
DampingEnergyInner: Reference
DampingEnergyOuter: Reference
Nsnapshots: 4
Nmonitor: 2
MonitorTimestep: 0.05
OutputDir: {out}
nbody:
- name: Star
  semi-major axis: 0.0 au
  mass: 1.0 solMass
- name: Jupiter
  semi-major axis: 1.0 au
  mass: 1.0e-3 solMass
  cubic smoothing factor: 0.6
"""


@pytest.mark.parametrize("eos,integrator,cooling", [("Isothermal", "Euler", "No"), ("Ideal", "Euler", "thermal"),
                                                    ("Ideal", "Leapfrog", "No")])
def test_restart_is_bitwise_identical(tmp_path, eos, integrator, cooling):
    """`fargocpt_hip restart 2 config.yml` and `auto` (restart_load, src/restart.cpp:18-139; start modes,
    src/start_mode.cpp:29-113): the run continued from snapshot 2 -- state, Q+ / Q-, the t = 0 grids of
    snapshots/reference for the damping zones, time, last dt and counters from misc.bin -- writes the same bits
    into snapshots 3 and 4 as the uninterrupted run."""
    import shutil
    out = tmp_path / "full"
    cfg = tmp_path / "config.yml"
    cfg.write_text(DISK_YML.format(eos=eos, integrator=integrator, cooling=cooling, out=out))
    r = subprocess.run([BIN, "-q", "start", str(cfg)], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr
    full = str(out) + "/"
    assert open(full + "snapshots/list.txt").read().split() == ["0", "1", "2", "3", "4"]
    assert os.path.exists(full + "snapshots/reference/Sigma.dat")
    fields = ["Sigma", "vrad", "vazi"] + (["energy", "Qplus", "Qminus"] if eos == "Ideal" else [])
    for mode in ("restart", "auto"):
        part = tmp_path / mode
        os.makedirs(part / "snapshots")
        keep = ["0", "1", "2", "reference"]
        for k in keep:
            shutil.copytree(full + "snapshots/" + k, part / "snapshots" / k)
        (part / "snapshots" / "list.txt").write_text("0\n1\n2\n")
        cfg2 = tmp_path / f"config_{mode}.yml"
        cfg2.write_text(DISK_YML.format(eos=eos, integrator=integrator, cooling=cooling, out=part))
        args = [BIN, "-q", "restart", "2", str(cfg2)] if mode == "restart" else [BIN, "-q", "auto", str(cfg2)]
        r = subprocess.run(args, capture_output=True, text=True, timeout=360)
        assert r.returncode == 0, r.stderr
        assert (part / "snapshots" / "list.txt").read_text().split() == ["0", "1", "2", "3", "4"]
        for n in ("3", "4"):
            a, b = _misc(full + f"snapshots/{n}/misc.bin"), _misc(str(part) + f"/snapshots/{n}/misc.bin")
            assert a == b, (mode, n, a, b)
            for f in fields:
                x = np.fromfile(full + f"snapshots/{n}/{f}.dat")
                y = np.fromfile(str(part) + f"/snapshots/{n}/{f}.dat")
                assert np.array_equal(x, y), (mode, n, f, np.abs(x - y).max())
    # `auto` on an empty output directory starts a fresh run
    fresh = tmp_path / "fresh"
    cfg3 = tmp_path / "config_fresh.yml"
    cfg3.write_text(DISK_YML.format(eos=eos, integrator=integrator, cooling=cooling, out=fresh))
    r = subprocess.run([BIN, "-q", "-N", "3", "auto", str(cfg3)], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr
    assert (fresh / "snapshots" / "0" / "Sigma.dat").exists()


def test_initial_state_from_2d_files(tmp_path):
    """SigmaCondition / EnergyCondition: 2D (init.cpp:1002-1007, 1307-1312): the grids of a snapshot as the
    initial state of a new run."""
    out = tmp_path / "a"
    cfg = tmp_path / "a.yml"
    cfg.write_text(DISK_YML.format(eos="Ideal", integrator="Euler", cooling="No", out=out))
    r = subprocess.run([BIN, "-q", "-N", "40", "start", str(cfg)], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr
    sig = np.fromfile(str(out) + "/snapshots/0/Sigma.dat") * 1.25
    en = np.fromfile(str(out) + "/snapshots/0/energy.dat") * 0.5
    sig.tofile(tmp_path / "sigma_in.dat")
    en.tofile(tmp_path / "energy_in.dat")
    out2 = tmp_path / "b"
    cfg2 = tmp_path / "b.yml"
    cfg2.write_text(DISK_YML.format(eos="Ideal", integrator="Euler", cooling="No", out=out2) +
                    f"SigmaCondition: 2D\nSigmaFilename: {tmp_path / 'sigma_in.dat'}\n"
                    f"EnergyCondition: 2D\nEnergyFilename: {tmp_path / 'energy_in.dat'}\n")
    r = subprocess.run([BIN, "-q", "-N", "5", "start", str(cfg2)], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(str(out2) + "/snapshots/0/Sigma.dat"), sig)
    assert np.array_equal(np.fromfile(str(out2) + "/snapshots/0/energy.dat"), en)
    cfg3 = tmp_path / "c.yml"
    cfg3.write_text(DISK_YML.format(eos="Ideal", integrator="Euler", cooling="No", out=tmp_path / "c") + "SigmaCondition: 1D\n")
    r = subprocess.run([BIN, "-q", "start", str(cfg3)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "not supported" in r.stderr


# ---- N ranks: the reference's `mpirun -np N fargocpt_exe start cfg.yml` ----------------------------------------------------
def _run_ranks(tmp_path, setup, outname, ranks, extra=(), edits=None, mode=("start",)):
    cfg = tmp_path / f"config_{outname}.yml"
    out = tmp_path / outname
    text = open(os.path.join(GOLDEN, "setups", setup)).read().splitlines()
    text = [("OutputDir: " + str(out)) if l.startswith("OutputDir") else l for l in text]
    for key, val in (edits or {}).items():
        text = [(f"{key}: {val}") if l.split(":")[0].strip() == key else l for l in text]
    cfg.write_text("\n".join(text) + "\n")
    cmd = [BIN, "-q"] + (["--ranks", str(ranks)] if ranks > 1 else []) + list(extra) + list(mode) + [str(cfg)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert not [p for p in os.listdir(out) if p.startswith(".fcpt_rdv")], "rendezvous directory left behind"
    return str(out) + "/"


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("ranks", [2, 3])
def test_n_rank_driver_matches_one_rank(tmp_path, ranks):
    """test/mpi_simple (128 x 384, examples/config.yml physics: locally isothermal disk + Jupiter, alpha viscosity, TW
    artificial viscosity, reflecting boundaries + damping) with `--ranks N`: N processes, one radial slab each, on the
    one GPU of the test box -- so the ghost rings and the CFL minimum travel through the library's host-staged
    transport (RCCL refuses two ranks on one device; with >= N GPUs the same command takes RCCL).  The reference's
    criterion is that snapshots/1/misc.bin exists (test/mpi_simple/check_results.py:6-11); here the files every slab
    wrote its window of (write2D, src/polargrid.cpp:135-180) must also equal the 1-process run's: same number of
    steps, Sigma / vrad / vazi to <= 1e-12 of max|field| (the reference itself: 4e-13 between 1 and 2 MPI ranks,
    BASELINE.md section 2)."""
    one = _run_ranks(tmp_path, "mpi_simple.yml", "one", 1)
    many = _run_ranks(tmp_path, "mpi_simple.yml", f"np{ranks}", ranks)
    assert os.path.exists(many + "snapshots/1/misc.bin")          # the reference's criterion
    m1, mn = _misc(one + "snapshots/1/misc.bin"), _misc(many + "snapshots/1/misc.bin")
    assert mn["n_iter"] == m1["n_iter"] and mn["n_iter"] > 20
    assert mn["time"] == pytest.approx(m1["time"], rel=1e-14) and mn["last_dt"] == pytest.approx(m1["last_dt"], rel=1e-10)
    assert open(many + "snapshots/list.txt").read().split() == ["0", "1"]
    assert np.array_equal(np.loadtxt(many + "used_rad.dat"), np.loadtxt(one + "used_rad.dat"))
    for snap in ("0", "reference", "1"):
        for name, rows in (("Sigma", 128), ("vrad", 129), ("vazi", 128)):
            a = np.fromfile(many + f"snapshots/{snap}/{name}.dat")
            b = np.fromfile(one + f"snapshots/{snap}/{name}.dat")
            assert a.size == b.size == rows * 384, (snap, name)
            if snap != "1":
                assert np.array_equal(a, b), (snap, name)          # the initial state: the same bits
            else:
                assert _rel(a, b) <= 1e-12, (snap, name, _rel(a, b))
                assert not np.array_equal(a, np.fromfile(one + f"snapshots/0/{name}.dat"))


def test_n_rank_restart_and_ideal_eos(tmp_path):
    """Two slabs with the energy equation (4 exchanged grids, Temperature / Qplus / Qminus windows) over two
    snapshots, then `restart 1` with two slabs: read2D's per-slab rows (src/polargrid.cpp:301-353) and restart_load
    (src/restart.cpp:18-139) reproduce the uninterrupted 2-slab run bit for bit, and both agree with one slab."""
    edits = {"EquationOfState": "Ideal", "Nsnapshots": "2", "MonitorTimestep": "0.314", "SurfaceCooling": "No"}
    one = _run_ranks(tmp_path, "mpi_simple.yml", "one", 1, edits=edits)
    two = _run_ranks(tmp_path, "mpi_simple.yml", "two", 2, edits=edits)
    names = ("Sigma", "vrad", "vazi", "energy", "Temperature", "Qplus", "Qminus")
    for name in names:
        a, b = np.fromfile(two + f"snapshots/2/{name}.dat"), np.fromfile(one + f"snapshots/2/{name}.dat")
        assert a.size == b.size
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= (1e-12 if name not in ("Qplus", "Qminus") else 1e-9) * scale, name
    full = {n: np.fromfile(two + f"snapshots/2/{n}.dat") for n in names}
    import shutil
    shutil.rmtree(two + "snapshots/2")
    with open(two + "snapshots/list.txt", "w") as f:
        f.write("0\n1\n")
    again = _run_ranks(tmp_path, "mpi_simple.yml", "two", 2, edits=edits, mode=("restart", "1"))
    assert _misc(again + "snapshots/2/misc.bin")["n_iter"] == _misc(one + "snapshots/2/misc.bin")["n_iter"]
    for n in names:
        assert np.array_equal(np.fromfile(again + f"snapshots/2/{n}.dat"), full[n]), n


def test_unknown_key_is_fatal_as_in_the_reference(tmp_path):
    """src/config.cpp:134-138: a key the reader does not know ends the run (a typo must not silently change the
    physics); --lenient downgrades it to a warning."""
    text = open(os.path.join(GOLDEN, "setups", "shocktube_SN.yml")).read()
    cfg = tmp_path / "typo.yml"
    cfg.write_text(text.replace("OutputDir", "OutputDir") + f"\nArtificalViscosity: TW\n")
    r = subprocess.run([BIN, "-q", "start", str(cfg)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "Unknown key(s) found in config file: 'artificalviscosity'" in r.stderr
