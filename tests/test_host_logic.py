"""Host-side logic of the product library (no GPU): ABI surface, descriptor defaults, radial
split, grid construction and initial conditions, checked against the oracle's restatement."""
import ctypes
import os
import re

import numpy as np
import pytest

import fargocpt_amd
from fargocpt_amd import binding as B, setups

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(product):
    hdr = open(os.path.join(ROOT, "include", "fargocpt_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(fcpt_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(product.cdll, s)]
    assert not missing, missing


def test_desc_struct_matches_header(product):
    d = product.desc_default()
    assert d.struct_size == ctypes.sizeof(B.Desc)
    assert d.abi_version == B.ABI_VERSION
    assert d.cfl == 0.5 and d.cfl_max_var == 1.1 and d.first_dt == 1e-9
    assert d.artificial_viscosity == B.ARTVISC_SN and d.fast_transport == 1
    assert d.G == 1.0 and d.Rgas == 1.0
    assert 1e4 < d.c_light < 1.1e4  # c in au / (yr / 2 pi)


def test_create_without_gpu_fails_loudly(product):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    d = setups.planet_disk(product, 32, 32)
    with pytest.raises(B.FcptError, match="FCPT_ENODEV"):
        product.create(d, product.radii(d))


def test_bad_descriptor_rejected(product):
    d = setups.planet_disk(product, 32, 32)
    d.struct_size = 8
    with pytest.raises(B.FcptError):
        product.create(d, np.zeros(64))


@pytest.mark.parametrize("nr,nranks", [(128, 1), (128, 2), (130, 4), (2048, 8)])
def test_split_domain(product, oracle, nr, nranks):
    """src/split.cpp:34-88"""
    covered = []
    for rank in range(nranks):
        d = setups.planet_disk(product, nr, 64)
        d.rank, d.nranks = rank, nranks
        s, so = product.split_domain(d), oracle.split_domain(d)
        for f, _ in B.Split._fields_:
            assert getattr(s, f) == getattr(so, f), f
        assert s.nr == s.imax - s.imin + 1
        lo = s.imin + s.zero_or_active
        hi = s.imin + s.max_or_active
        covered.append((lo, hi))
        if nranks == 1:
            assert (s.zero_no_ghost, s.one_no_ghost_vr, s.max_no_ghost, s.maxmo_no_ghost_vr) == (1, 2, nr - 1, nr - 1)
            assert (s.radial_first_active, s.radial_active_size) == (1, nr - 1)
    assert covered[0][0] == 0 and covered[-1][1] == nr
    for (a, b), (c, e) in zip(covered, covered[1:]):
        assert b == c  # write windows tile the global grid (polargrid.cpp:150-172)


def test_split_too_narrow(product):
    d = setups.planet_disk(product, 40, 64)
    d.rank, d.nranks = 0, 4  # 10 rings per slab < 2 * CPUOVERLAP
    with pytest.raises(B.FcptError, match="FCPT_ESPLIT"):
        product.split_domain(d)


@pytest.mark.parametrize("spacing", [B.SPACING_ARITHMETIC, B.SPACING_LOGARITHMIC, B.SPACING_EXPONENTIAL])
def test_radii(product, oracle, spacing):
    """src/init.cpp:92-145: Radii[1] = Rmin, Radii[N-1] = Rmax, one ghost cell outside each."""
    d = setups.planet_disk(product, 64, 32)
    d.radial_spacing = spacing
    r, ro = product.radii(d), oracle.radii(d)
    assert np.array_equal(r, ro)
    assert r[1] == pytest.approx(d.rmin, rel=1e-15) and r[d.nr_global - 1] == pytest.approx(d.rmax, rel=1e-13)
    assert np.all(np.diff(r) > 0)


def _cutoff_disk(product, adiabatic):
    d = setups.planet_disk(product, 48, 16, adiabatic=adiabatic)
    d.profile_cutoff_outer, d.profile_cutoff_point_outer, d.profile_cutoff_width_outer = 1, 1.8, 0.1
    d.profile_cutoff_inner, d.profile_cutoff_point_inner, d.profile_cutoff_width_inner = 1, 0.6, 0.05
    d.set_sigma0, d.disk_mass = 1, 0.01
    return d


@pytest.mark.parametrize("name", ["planet_iso", "planet_adi", "ring", "shock", "cutoff_iso", "cutoff_adi"])
def test_initial_fields_match_oracle(product, oracle, name):
    d = {"planet_iso": lambda: setups.planet_disk(product, 48, 16),
         "planet_adi": lambda: setups.planet_disk(product, 48, 16, adiabatic=True),
         "ring": lambda: setups.spreading_ring(product, 64, 4),
         "shock": lambda: setups.shocktube(product, 64, 4),
         "cutoff_iso": lambda: _cutoff_disk(product, False),
         "cutoff_adi": lambda: _cutoff_disk(product, True)}[name]()
    radii = product.radii(d)
    d1, d2 = d.copy(), d.copy()
    f1, f2 = product.initial_fields(d1, radii), oracle.initial_fields(d2, radii)
    assert d1.sigma0 == pytest.approx(d2.sigma0, rel=1e-13)
    for a, b, n in zip(f1, f2, ("sigma", "vrad", "vazi", "energy")):
        # std::cyl_bessel_i vs the oracle's series differ at the 1e-15 level for the ring
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=0, err_msg=n)
    if name == "shock":
        assert set(np.unique(f1[0])) == {0.125, 1.0}
    if name == "ring":
        assert abs(d1.sigma0 / d.sigma0 - 1) < 0.05  # SetSigma0 renormalisation (init.cpp:1150-1185)
    if name.startswith("cutoff"):
        # ProfileCutoffInner / Outer (init.cpp:1063-1146, util.cpp:69-93): the power law times two logistic edges,
        # then SetSigma0 scales the cut profile to the disk mass
        ri = np.asarray(radii[:d.nr_global + 1])
        rm = 2.0 / 3.0 * (ri[1:] ** 3 - ri[:-1] ** 3) / (ri[1:] ** 2 - ri[:-1] ** 2)
        shape = rm ** -0.5 / (1 + np.exp((rm - 1.8) / 0.1)) / (1 + np.exp((0.6 - rm) / 0.05))
        sig = f1[0][:, 0]
        np.testing.assert_allclose(sig / sig[24], shape / shape[24], rtol=1e-12)
        surf = np.pi * (ri[1:] ** 2 - ri[:-1] ** 2)
        assert np.sum(surf[1:-1] * sig[1:-1]) == pytest.approx(0.01, rel=1e-12)
        if name == "cutoff_adi":    # e = Sigma0 h^2 r^(-SigmaSlope - 1 + 2 FlaringIndex) / (gamma - 1) times the same edges
            e = f1[3][:, 0]
            eshape = shape / rm
            np.testing.assert_allclose(e / e[24], eshape / eshape[24], rtol=1e-12)


def test_kernel_name_table(product):
    names = product.kernel_names()
    assert "k_transport_radial" in names and len(set(names)) == len(names)


def test_hot_kernel_occupancy_budget():
    """The marching kernels are bound by the vector ALUs and by latency at 3-4 wavefronts per SIMD: their
    register budget is part of the design.  Cross-compiles the kernels for gfx950
    with -Rpass-analysis=kernel-resource-usage and checks occupancy and spills of the bench path."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fargocpt_amd", "csrc", "fcpt_kernels.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull, src],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    usage, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
        if m and name:
            usage[name][m.group(1).strip()] = int(m.group(2))
    budget = {  # mangled-name fragment -> (minimum waves per SIMD, scratch bytes per lane allowed)
        "17k_transport_fusedILi1ELb0ELb1ELi0E": (4, 0),   # isothermal, damping folded in, van Leer: the bench kernel
        # ideal EOS: 4 waves since round 2 (register diet + waves_per_eu); the instantiation that also stores the
        # cell-local CFL terms parks up to seven dwords in scratch
        # (round 3: one value of the rare in-launch radial sweep -- taken when k_ring_mean raised the shift-jump flag --
        #  is parked in scratch across its grid-stride loop: one store and one reload per wavefront OF THAT PATH, none
        #  in the marching loop)
        "17k_transport_fusedILi1ELb1ELb1ELi0E": (4, 16),
        "23k_transport_fused_thermILi1ELb1ELb1ELi0E": (4, 32),
        "14k_source_marchILi1ELb0ELb0E": (6, 0),            # isothermal source step, TW artificial viscosity
        "14k_source_marchILi1ELb1ELb0E": (4, 0),            # ... with StabilizeViscosity
        # ideal EOS, potential in the kernel, no cooling terms: one loop-invariant pair lives in scratch since the pair
        # reciprocals are carried through the window (measured: -1.4 % per step with it)
        "18k_source_march_adiILi1ELb1E": (4, 16),
        "k_cfl_ringsILb0ELi8E": (6, 0),                   # Nphi <= 4096
        "k_cfl_ringsILb0ELi16E": (4, 0),                  # rings of up to 8192 cells
    }
    for frag, (waves, scratch) in budget.items():
        hits = [k for k in usage if frag in k]
        assert hits, frag
        u = usage[hits[0]]
        assert u["Occupancy [waves/SIMD]"] >= waves, (frag, u)
        assert u["ScratchSize [bytes/lane]"] <= scratch, (frag, u)
        if scratch == 0:
            assert u["VGPRs Spill"] == 0, (frag, u)


def test_nan_radii_of_a_coarse_exponential_grid_are_refused(product, oracle):
    """The reference's Newton iteration for `RadialSpacing: Exponential` (src/init.cpp:113-131) collapses to the
    trivial root for coarse grids and fills Radii with NaN; both libraries refuse such a grid at create time
    (before any device is touched) instead of marching NaNs."""
    from fargocpt_amd import driver, setups
    d = setups.planet_disk(product, 16, 64)
    d.radial_spacing = B.SPACING_EXPONENTIAL
    radii = product.radii(d)
    assert not np.isfinite(radii).all()
    assert np.array_equal(np.isnan(radii), np.isnan(oracle.radii(d)))
    for lib in (oracle, product):
        with pytest.raises(B.FcptError, match="FCPT_EINVAL"):
            driver.make_context(lib, d)


def test_option_names_are_documented():
    """Every kernel-selection switch of the library (FCPT_OPTION_NAMES in csrc/fcpt_internal.h) is named in the public
    header's description of fcpt_set_option, and nothing on the launch path reads the environment."""
    internal = open(os.path.join(ROOT, "fargocpt_amd", "csrc", "fcpt_internal.h")).read()
    block = internal[internal.index("#define FCPT_OPTION_NAMES"):]
    block = block[:block.index("\n\n")]
    names = re.findall(r"X\((\w+)\)", block)
    assert len(names) >= 15 and len(set(names)) == len(names)
    hdr = open(os.path.join(ROOT, "include", "fargocpt_hip.h")).read()
    doc = hdr[hdr.index("Kernel-selection switches of one context"):hdr.index("int fcpt_set_option")]
    missing = [n for n in names if not re.search(r"\b%s\b" % n, doc)]
    assert not missing, missing
    for f in ("kernels/launch.h", "fcpt_step.hip", "fcpt_exchange.hip"):
        assert "getenv" not in open(os.path.join(ROOT, "fargocpt_amd", "csrc", f)).read(), f


def test_rank_launcher_of_the_host_driver_reports_and_cleans_up(tmp_path):
    """`fargocpt_hip --ranks 2` on a box without a GPU: both ranks stop with "no HIP device" (there is no CPU path),
    the parent -- which started them before touching any GPU API -- reports the first failure, ends the other rank,
    returns non-zero and leaves no rendezvous directory behind.  (With a GPU the same command runs: tests/test_gpu_driver.py.)"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the launcher is exercised by tests/test_gpu_driver.py")
    exe = os.path.join(ROOT, "fargocpt_amd", "bin", "fargocpt_hip")
    cfg = tmp_path / "c.yml"
    cfg.write_text(f"Nrad: 64\nNaz: 64\nOutputDir: {tmp_path}/out\n")
    r = subprocess.run([exe, "-q", "--ranks", "2", "start", str(cfg)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert "no HIP device" in r.stderr and "ending the other ranks" in r.stderr
    assert not [p for p in os.listdir(tmp_path / "out") if p.startswith(".fcpt_rdv")]
    # a key the reference's reader does not know is fatal before any rank starts (src/config.cpp:134-138)
    cfg.write_text(f"Nrad: 64\nNazz: 64\nOutputDir: {tmp_path}/out\n")
    r = subprocess.run([exe, "--ranks", "2", "start", str(cfg)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "Unknown key(s) found in config file: 'nazz'" in r.stderr and "no HIP device" not in r.stderr


@pytest.mark.parametrize("nr,nphi,adiabatic,damp", [(2048, 4096, False, 100), (2048, 4096, True, 100), (2048, 6144, False, 0),
                                                    (1024, 3072, True, 50), (4096, 4096, False, 0), (1367, 2731 * 2, True, 33),
                                                    (512, 1536, False, 20), (128, 384, False, 0), (100, 2, False, 0)])
def test_chunk_tables_of_the_marching_kernels(product, nr, nphi, adiabatic, damp):
    """transport_schedule / source_schedule (kernels/launch.h) as host logic for an MI355X (256 CUs): whatever the
    grid, a table that is handed to the kernels holds every ring (every (segment, ring)) exactly once, no chunk is
    empty, the long chunks come first; small grids get no table (equal chunks)."""
    t, s = product.selftest_chunk_tables(nr, nphi, 256, adiabatic, damp, damp)
    if len(t):
        tiles = -(-nphi // 53)
        assert len(t) % 4 == 0
        live = t[t[:, 2] > t[:, 1]]
        cover = np.zeros((tiles, nr), dtype=np.int32)
        for tl, a, b in live:
            cover[tl, a:b] += 1
        assert (cover == 1).all()
        n = live[:, 2] - live[:, 1]
        assert n.min() >= 1
        if len(live) > 256 * 4 * 4:                     # several rounds of wavefronts: graded chunks, long ones first
            c0 = live[live[:, 0] == 0]                  # tile 0 of every chunk, in dispatch order
            m = c0[:, 2] - c0[:, 1]
            k0 = 8 * -(-512 // tiles)                   # 8 * ceil(slots per XCD / tiles): the first level
            assert len(m) > k0 and m[k0 - 8:k0].min() >= 2 * m[-9:-1].max()
            if damp:                                    # a damping-zone ring counts 1.4: those chunks hold fewer rings
                assert m[0] < m[k0 - 1]
        else:                                           # one round: the first wavefronts an XCD receives march the longest chunks
            x0 = t[(np.arange(len(t)) // 4) % 8 == 0]
            x0 = x0[x0[:, 2] > x0[:, 1]]
            mm = x0[:, 2] - x0[:, 1]
            assert mm[:32].mean() > mm[-32:].mean() and n.min() >= 2
    if len(s):
        segs = -(-nphi // 59)
        assert len(s) % 32 == 0                         # whole workgroups, the same number for each of the 8 XCDs
        live = s[s[:, 2] > s[:, 1]]
        cover = np.zeros((segs, nr + 1), dtype=np.int32)
        for sg, a, b in live:
            cover[sg, a:b] += 1
        assert (cover == 1).all()
        assert (live[:, 2] - live[:, 1]).min() >= 3     # the boundary call folded into the kick needs three rows
        per_xcd = len(s) // 8
        assert len(live) <= 256 * 4 * (4 if adiabatic else 6)   # one round of wavefronts
        x0 = s[(np.arange(len(s)) // 4) % 8 == 0]
        x0 = x0[x0[:, 2] > x0[:, 1]]
        m = x0[:, 2] - x0[:, 1]
        assert m[:32].mean() > m[-32:].mean()
        assert per_xcd * 8 == len(s)
    if nr * nphi <= 512 * 1536:
        assert len(t) == 0 and len(s) == 0
    if (nr, nphi) == (1024, 3072):
        assert 0 < len(t[t[:, 2] > t[:, 1]]) <= 4096    # config 3: one round, rank-matched
