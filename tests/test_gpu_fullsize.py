"""BASELINE.json's full sizes (512 x 1536, 1024 x 3072, 2048 x 6144 in four slabs, 4096 x 4 and the headline
2048 x 4096).  Two kinds of checks:

1. HIP against the oracle, field by field (1e-10, dt histories 1e-9), over a handful of steps: the oracle does
   35-46 M cell-updates/s on the GPU box's host cores, i.e. a few seconds per case (tests `*_against_oracle`);
2. size-independent properties of the path over more steps than the oracle is worth waiting for:

* azimuthal symmetry: an axisymmetric disk without planet stays axisymmetric to rounding -- every
  wavefront tile seam of the marching kernels and every integer shift of the FARGO transport would
  show up as a phi-dependence;
* rotation equivariance: rotating the initial perturbation by k cells rotates the result by k cells
  (the tiles then cut the data elsewhere);
* conservation: closed (reflecting) boundaries, no damping: the transport is in flux form, the disk
  mass is conserved to rounding, and the angular momentum changes only through the viscous torque at the
  walls (bounded far below the planet-free advection terms);
* slab independence: 2 and 4 radial slabs with ghost exchange agree with the single slab (the reference
  itself: 4e-13 between 1 and 2 ranks);
* the device-resident time-step loop equals the host-driven loop."""
import numpy as np
import pytest

from fargocpt_amd import binding as B, driver, setups
from tests.util import perturb, rel_err, run_pair

pytestmark = pytest.mark.gpu
NR, NPHI = 2048, 4096
TOL, TOL_DT = 1e-10, 1e-9


def _check_pair(res, fields):
    (sa, dta), (sb, dtb) = res
    assert np.allclose(dta, dtb, rtol=TOL_DT, atol=0), (dta, dtb)
    errs = {k: rel_err(sa[k], sb[k]) for k in fields}
    assert all(np.isfinite(sa[k]).all() for k in fields)
    assert max(errs.values()) <= TOL, errs
    return errs


# BASELINE.json configs 2, 3, 4 and the headline workload at their own sizes: (name, nr, nphi, ideal EOS, HIP slabs, steps)
FULL_CASES = [
    ("config2_512x1536", 512, 1536, False, 1, 10),
    ("config3_1024x3072_ideal", 1024, 3072, True, 1, 8),
    ("config4_2048x6144_4slabs", 2048, 6144, False, 4, 5),
    ("headline_2048x4096", 2048, 4096, False, 1, 6),
    ("headline_2048x4096_ideal", 2048, 4096, True, 1, 5),
]


@pytest.mark.parametrize("name,nr,nphi,adiabatic,nslabs,nsteps", FULL_CASES, ids=[c[0] for c in FULL_CASES])
def test_full_size_against_oracle(product, oracle, name, nr, nphi, adiabatic, nslabs, nsteps):
    """The BASELINE configurations at full size, HIP (in `nslabs` radial slabs) against the single-slab oracle."""
    d = setups.planet_disk(product, nr, nphi, adiabatic=adiabatic)
    fields = ("sigma", "vrad", "vazi") + (("energy",) if adiabatic else ())
    _check_pair(run_pair(product, oracle, d, nsteps, bodies=setups.jupiter_bodies(d), nslabs=(nslabs, 1)), fields)


# (configs 1 and 5 -- 128 x 384 and 4096 x 4 -- run at their BASELINE sizes in tests/test_gpu_parity.py)


@pytest.mark.parametrize("adiabatic", [False, True])
def test_fallback_transport_at_full_size_against_oracle(product, oracle, adiabatic):
    """A step of 4x the CFL time step at 2048 x 4096: k_transport_fused meets |Nshift[i] - Nshift[i-1]| > 1 and
    hands over to the two-kernel transport queued behind it, whose grid is capped at 256 blocks with a grid-stride
    loop -- the form that only differs from the small-grid one above 256 virtual blocks."""
    d = setups.planet_disk(product, NR, NPHI, adiabatic=adiabatic)
    d.damping = 0
    d.first_dt = 1.0  # no 1.1x ramp: the first step already runs at the CFL limit
    fields = ("sigma", "vrad", "vazi") + (("energy",) if adiabatic else ())
    _check_pair(run_pair(product, oracle, d, 3, bodies=setups.jupiter_bodies(d), dt_scale=4.0), fields)


def _run(lib, d, fields, radii, nsteps, bodies=None, nslabs=1, device_loop=False):
    ctxs = []
    for rank in range(nslabs):
        dd = d.copy()
        dd.rank, dd.nranks = rank, nslabs
        s = lib.split_domain(dd)
        sub = tuple(np.ascontiguousarray(f[s.imin:s.imin + s.nr + (1 if k == 1 else 0)]) for k, f in enumerate(fields))
        ctxs.append(driver.make_context(lib, dd, fields=sub, radii=radii, bodies=bodies))
    S = driver.SlabSet(ctxs)
    S.prepare()
    if device_loop:
        assert nslabs == 1
        ctxs[0].run_steps(nsteps, snap=False)
        dts = None
    else:
        dts = S.run(nsteps)
    out = S.gather()
    t = ctxs[0].clock.time
    for c in ctxs:
        c.close()
    return out, dts, t


@pytest.fixture(scope="module")
def base(product):
    d = setups.planet_disk(product, NR, NPHI)
    d0 = d.copy()
    radii = product.radii(d0)
    fields = product.initial_fields(d0, radii)
    return d0, radii, fields


def test_axisymmetric_disk_stays_axisymmetric(product, base):
    d, radii, fields = base
    out, _, _ = _run(product, d, fields, radii, 30)
    for k in ("sigma", "vrad", "vazi"):
        a = out[k]
        scale = np.abs(out["vazi"]).max() if k == "vrad" else np.abs(a).max()  # v_r itself is ~1e-5 v_phi
        spread = np.abs(a - a[:, :1]).max() / scale
        # the star's potential r^2 (cos^2 + sin^2) carries rounding noise in phi, which the upwind
        # switches amplify a little; a seam or shift error would be of order 1e-3
        assert spread <= 1e-11, (k, spread)


def test_rotation_equivariance_and_device_loop(product, base):
    d, radii, fields = base
    bodies = None  # no planet: the problem is invariant under rotations by whole cells
    f0 = perturb(fields, d, 1e-3)
    ref, _, t_ref = _run(product, d, f0, radii, 20, bodies)
    k = 1237  # not a multiple of any tile stride
    f1 = tuple(np.roll(f, k, axis=1) for f in f0)
    rot, _, _ = _run(product, d, f1, radii, 20, bodies)
    for name in ("sigma", "vrad", "vazi"):
        # (same phi-dependent rounding of the potential as above; v_r on the scale of v_phi)
        scale = np.abs(ref["vazi"]).max() if name == "vrad" else np.abs(ref[name]).max()
        assert np.abs(np.roll(rot[name], -k, axis=1) - ref[name]).max() / scale <= 1e-11, name
    dev, _, t_dev = _run(product, d, f0, radii, 20, bodies, device_loop=True)
    assert t_dev == t_ref
    for name in ("sigma", "vrad", "vazi"):
        assert np.array_equal(dev[name], ref[name]), name


@pytest.mark.parametrize("adiabatic", [False, True])
def test_mass_and_angular_momentum_closed_box(product, adiabatic):
    d = setups.planet_disk(product, NR, NPHI, adiabatic=adiabatic, damping=False)
    d0 = d.copy()
    radii = product.radii(d0)
    fields = perturb(product.initial_fields(d0, radii), d0, 1e-3)
    ri, rs = radii[:NR], radii[1:NR + 1]
    surf = np.pi * (rs ** 2 - ri ** 2) / NPHI
    rmed = 2.0 / 3.0 * (rs ** 3 - ri ** 3) / (rs ** 2 - ri ** 2)

    def totals(st):
        # the closed box: rings 1 .. Nr-2 between the reflecting interfaces 1 and Nr-1 (the ghost rings
        # 0 and Nr-1 are overwritten by the zero-gradient condition every step)
        a = slice(1, NR - 1)
        m = (st["sigma"][a] * surf[a, None]).sum()
        l = (st["sigma"][a] * surf[a, None] * rmed[a, None] * (st["vazi"][a] + rmed[a, None] * d0.omega_frame)).sum()
        return m, l

    m0, l0 = totals({"sigma": fields[0], "vazi": fields[2]})
    out, _, _ = _run(product, d0, fields, radii, 30, setups.jupiter_bodies(d0))
    m1, l1 = totals(out)
    assert abs(m1 / m0 - 1.0) <= 1e-13
    # a Jupiter-mass planet exchanges angular momentum with the disk: a few 1e-8 of L per step at most
    assert abs(l1 / l0 - 1.0) <= 1e-5


@pytest.mark.parametrize("nphi,nslabs,adiabatic", [(4096, 2, False), (6144, 4, False), (3072, 2, True)])
def test_radial_slabs_at_full_size(product, nphi, nslabs, adiabatic):
    d = setups.planet_disk(product, NR if not adiabatic else 1024, nphi, adiabatic=adiabatic)
    d0 = d.copy()
    radii = product.radii(d0)
    fields = perturb(product.initial_fields(d0, radii), d0, 1e-3)
    bodies = setups.jupiter_bodies(d0)
    one, dt1, _ = _run(product, d0, fields, radii, 12, bodies, 1)
    many, dtn, _ = _run(product, d0, fields, radii, 12, bodies, nslabs)
    assert np.allclose(dt1, dtn, rtol=1e-12, atol=0)
    for k in ("sigma", "vrad", "vazi") + (("energy",) if adiabatic else ()):
        assert rel_err(many[k], one[k]) <= 1e-12, k


@pytest.mark.parametrize("adiabatic", [False, True])
def test_bench_loop_is_reproducible_bit_for_bit(product, adiabatic):
    """Every reduction of the path runs in a fixed order: the bench workload (device-resident loop, 40 steps)
    run twice gives identical bits.  A race between wavefronts -- the failure mode the small grids of the parity
    tests are least likely to show -- would not."""
    d = setups.planet_disk(product, NR, NPHI if not adiabatic else 3072, adiabatic=adiabatic)
    d0 = d.copy()
    radii = product.radii(d0)
    fields = perturb(product.initial_fields(d0, radii), d0, 1e-3)
    bodies = setups.jupiter_bodies(d0)
    a, _, ta = _run(product, d0, fields, radii, 40, bodies=bodies, device_loop=True)
    b, _, tb = _run(product, d0, fields, radii, 40, bodies=bodies, device_loop=True)
    assert ta == tb
    for k in ("sigma", "vrad", "vazi") + (("energy",) if adiabatic else ()):
        assert np.array_equal(a[k], b[k]), k
    assert all(np.isfinite(v).all() for v in a.values())


@pytest.mark.parametrize("adiabatic", [False, True])
def test_graded_transport_chunks_at_full_size(product, adiabatic):
    """At the bench size the fused transport needs two rounds of the GPU's wavefront slots, and the library grades its
    chunk lengths (transport_schedule in kernels/launch.h): the table must hold every ring exactly once, start with
    the same number of long chunks for each of the 8 XCDs, end with short ones -- and give the bits of equal chunks."""
    d = setups.planet_disk(product, NR, NPHI, adiabatic=adiabatic)
    d0 = d.copy()
    radii = product.radii(d0)
    fields = perturb(product.initial_fields(d0, radii), d0, 1e-3)
    bodies = setups.jupiter_bodies(d0)
    out = []
    for graded in (1, 0):
        ctx = driver.make_context(product, d0, fields=fields, radii=radii, bodies=bodies)
        ctx.set_option("transport_graded", graded)
        tab = ctx.transport_chunks()
        if graded:
            tiles = -(-NPHI // 53)
            live = tab[tab[:, 2] > tab[:, 1]]
            assert len(live) > 16 * tiles
            cover = np.zeros((tiles, NR), dtype=np.int32)
            for tl, a, b in live:
                cover[tl, a:b] += 1
            assert (cover == 1).all()
            n = live[:, 2] - live[:, 1]
            assert n[:64].min() > 2 * n[-64:].max()
            assert ctx.get_option("fused_damping") == 1
            # tile 0 of every chunk, in dispatch order: the chunks themselves
            c0 = live[live[:, 0] == 0]
            m = c0[:, 2] - c0[:, 1]
            big = m[16]                                 # (the first chunks sit in the damping zones: fewer, costlier rings)
            n_long = int(np.argmax(m < 0.6 * big))      # chunks before the first markedly shorter one
            assert n_long >= 8 and n_long % 8 == 0, (n_long, m[:80])
        else:
            assert len(tab) == 0
        S = driver.SlabSet([ctx])
        S.prepare()
        assert ctx.run_steps(12) == 12
        out.append((ctx.state(), ctx.clock.time))
        ctx.close()
    assert out[0][1] == out[1][1]
    for k in out[0][0]:
        assert np.array_equal(out[0][0][k], out[1][0][k]), k


@pytest.mark.parametrize("adiabatic", [False, True])
def test_rank_matched_source_chunks_at_full_size(product, adiabatic):
    """The marching source kernels run as one round of wavefronts at the bench size; the library matches every
    wavefront's chunk length to the rank at which its SIMD will serve it (source_schedule in kernels/launch.h).  The
    table must hold every (segment, ring) exactly once, give the first wavefronts of an XCD longer chunks than its last,
    and give the bits of equal chunks (every ring of a segment is computed by one wavefront from the same operands; the
    ring sums of v_phi are per segment)."""
    d = setups.planet_disk(product, NR, NPHI, adiabatic=adiabatic)
    d0 = d.copy()
    radii = product.radii(d0)
    fields = perturb(product.initial_fields(d0, radii), d0, 1e-3)
    bodies = setups.jupiter_bodies(d0)
    out = []
    for graded in (-1, 0):
        ctx = driver.make_context(product, d0, fields=fields, radii=radii, bodies=bodies)
        ctx.set_option("source_graded", graded)
        tab = ctx.source_chunks()
        if graded:
            segs = (NPHI + 58) // 59
            assert len(tab) % 4 == 0 and len(tab) >= segs * 8
            live = tab[tab[:, 2] > tab[:, 1]]
            cover = np.zeros((segs, NR + 1), dtype=np.int32)
            for sg, k0, k1 in live:
                cover[sg, k0:k1] += 1
            assert (cover == 1).all()
            n = live[:, 2] - live[:, 1]
            assert n.min() >= 3
            xcd0 = tab[(np.arange(len(tab)) // 4) % 8 == 0]          # workgroups 0, 8, 16, ... = dispatch order of XCD 0
            xcd0 = xcd0[xcd0[:, 2] > xcd0[:, 1]]
            m = xcd0[:, 2] - xcd0[:, 1]
            assert m[:64].mean() > 1.1 * m[-64:].mean()
        else:
            assert len(tab) == 0
        S = driver.SlabSet([ctx])
        S.prepare()
        assert ctx.run_steps(12) == 12
        out.append((ctx.state(), ctx.clock.time))
        ctx.close()
    assert out[0][1] == out[1][1]
    for k in out[0][0]:
        assert np.array_equal(out[0][0][k], out[1][0][k]), k
