"""ctypes mirror of include/fargocpt_hip.h.

Plumbing only: structures, error handling and a thin object wrapper over the
C ABI.  The same wrapper can bind any library that exports the ABI under a
symbol prefix (`fcpt_` for the product); nothing here computes anything.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

ABI_VERSION = 5
OVERLAP = 7
GEOM_PAD = 15
MAX_BODIES = 8

# enum values (include/fargocpt_hip.h)
SPACING_ARITHMETIC, SPACING_LOGARITHMIC, SPACING_EXPONENTIAL = 0, 1, 2
EOS_ISOTHERMAL, EOS_IDEAL = 0, 1
ARTVISC_NONE, ARTVISC_TW, ARTVISC_SN = 0, 1, 2
OPACITY_LIN, OPACITY_BELL, OPACITY_CONST, OPACITY_SIMPLE = 0, 1, 2, 3
BETAREF_ZERO, BETAREF_REFERENCE, BETAREF_MODEL, BETAREF_FLOOR = 0, 1, 2, 3
LIMITER_VANLEER, LIMITER_MC = 0, 1
INTEGRATOR_EULER, INTEGRATOR_LEAPFROG = 0, 1
(BC_ZEROGRADIENT, BC_REFERENCE, BC_REFLECTING, BC_OUTFLOW, BC_KEPLERIAN, BC_ZEROSHEAR,
 BC_NONE) = range(7)
DAMP_NONE, DAMP_REFERENCE, DAMP_ZERO, DAMP_MEAN = range(4)
IC_PROFILE, IC_SPREADING_RING, IC_SHOCKTUBE = range(3)
(F_SIGMA, F_VRAD, F_VAZI, F_ENERGY, F_PRESSURE, F_SOUNDSPEED, F_SCALE_HEIGHT, F_VISCOSITY,
 F_TEMPERATURE, F_POTENTIAL, F_SIGMA0, F_VRAD0, F_VAZI0, F_ENERGY0, F_QPLUS, F_QMINUS,
 F_VISC_CFAC_PHI, F_VISC_CFAC_R, F_MASSFLOW, F_ACCEL_RADIAL, F_ACCEL_AZIMUTHAL) = range(21)
VECTOR_FIELDS = (F_VRAD, F_VRAD0, F_MASSFLOW, F_ACCEL_RADIAL, F_ACCEL_AZIMUTHAL)
COMM_ID_BYTES = 128

ERRORS = {-1: "FCPT_EINVAL", -2: "FCPT_ENOMEM", -3: "FCPT_EHIP", -4: "FCPT_ESPLIT", -5: "FCPT_ENODEV", -6: "FCPT_ESHEAR",
          -7: "FCPT_ECOMM"}

_i32, _u32, _u64, _f64 = C.c_int32, C.c_uint32, C.c_uint64, C.c_double


class Desc(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("abi_version", _u32),
        ("nr_global", _i32), ("nphi", _i32), ("rank", _i32), ("nranks", _i32),
        ("radial_spacing", _i32), ("_pad0", _i32),
        ("rmin", _f64), ("rmax", _f64), ("exponential_cell_size_factor", _f64),
        ("eos", _i32), ("_pad1", _i32),
        ("adiabatic_index", _f64), ("mu", _f64), ("aspect_ratio", _f64), ("flaring_index", _f64),
        ("minimum_temperature", _f64), ("maximum_temperature", _f64),
        ("sigma0", _f64), ("sigma_slope", _f64), ("sigma_floor", _f64),
        ("viscous_alpha", _f64), ("constant_viscosity", _f64), ("radial_viscosity_factor", _f64),
        ("stabilize_viscosity", _i32), ("artificial_viscosity", _i32),
        ("artificial_viscosity_factor", _f64),
        ("artificial_viscosity_dissipation", _i32), ("heating_viscous", _i32),
        ("heating_viscous_factor", _f64),
        ("fast_transport", _i32), ("flux_limiter", _i32),
        ("integrator", _i32), ("_pad2", _i32),
        ("cfl", _f64), ("cfl_max_var", _f64), ("first_dt", _f64),
        ("heating_cooling_cfl_limit", _f64), ("monitor_timestep", _f64),
        ("nmonitor", _i32), ("nsnapshots", _i32),
        ("omega_frame", _f64), ("thickness_smoothing", _f64),
        ("body_force_from_potential", _i32), ("_pad3", _i32),
        ("hydro_center_mass", _f64),
        ("bc_sigma", _i32 * 2), ("bc_energy", _i32 * 2), ("bc_vrad", _i32 * 2), ("bc_vaz", _i32 * 2),
        ("keplerian_vaz_factor", _f64 * 2), ("keplerian_vrad_factor", _f64 * 2),
        ("damping", _i32), ("_pad4", _i32),
        ("damping_inner_limit", _f64), ("damping_outer_limit", _f64),
        ("damping_time_factor", _f64), ("damping_time_radius_outer", _f64),
        ("damp_vrad", _i32 * 2), ("damp_vaz", _i32 * 2), ("damp_sigma", _i32 * 2),
        ("damp_energy", _i32 * 2),
        ("G", _f64), ("Rgas", _f64), ("sigma_sb", _f64), ("c_light", _f64),
        ("ic", _i32), ("set_sigma0", _i32), ("disk_mass", _f64),
        ("initialize_vradial_zero", _i32), ("initialize_pure_keplerian", _i32),
        ("cooling_surface", _i32), ("opacity", _i32),
        ("cooling_radiative_factor", _f64), ("kappa_const", _f64), ("kappa_factor", _f64),
        ("tau_factor", _f64), ("tau_min", _f64), ("density_factor", _f64),
        ("cooling_beta", _i32), ("cooling_beta_reference", _i32),
        ("cooling_beta_value", _f64), ("cooling_beta_ramp_up", _f64),
        ("temperature_cgs", _f64), ("density_cgs", _f64), ("opacity_cgs", _f64),
        ("profile_cutoff_inner", _i32), ("profile_cutoff_outer", _i32),
        ("profile_cutoff_point_inner", _f64), ("profile_cutoff_width_inner", _f64),
        ("profile_cutoff_point_outer", _f64), ("profile_cutoff_width_outer", _f64),
        ("write_massflow", _i32), ("_pad5", _i32),
    ]

    def copy(self) -> "Desc":
        d = Desc()
        C.memmove(C.byref(d), C.byref(self), C.sizeof(Desc))
        return d


class Split(C.Structure):
    _fields_ = [(n, _i32) for n in (
        "nr", "imin", "imax", "zero_no_ghost", "one_no_ghost_vr", "max_no_ghost",
        "maxmo_no_ghost_vr", "zero_or_active", "max_or_active", "radial_first_active",
        "radial_active_size", "is_first", "is_last")]


class Clock(C.Structure):
    _fields_ = [("time", _f64), ("last_dt", _f64), ("n_hydro_iter", _u64),
                ("n_monitor", _u32), ("n_snapshot", _u32)]


class FcptError(RuntimeError):
    pass


_dp = C.POINTER(_f64)


def _as_dp(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


class Library:
    """One loaded shared object exporting the ABI under `prefix`."""

    def __init__(self, cdll: C.CDLL, prefix: str = "fcpt_"):
        self.cdll = cdll
        self.prefix = prefix

    def fn(self, name: str):
        return getattr(self.cdll, self.prefix + name)

    def has(self, name: str) -> bool:
        return hasattr(self.cdll, self.prefix + name)

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = ""
            if self.has("last_error"):
                f = self.fn("last_error")
                f.restype = C.c_char_p
                m = f()
                msg = (": " + m.decode()) if m else ""
            raise FcptError(f"{self.prefix}{what} failed with {ERRORS.get(rc, rc)}{msg}")

    # ---- host helpers -------------------------------------------------------
    def desc_default(self) -> Desc:
        d = Desc()
        self.check(self.fn("desc_default")(C.byref(d)), "desc_default")
        return d

    def split_domain(self, d: Desc) -> Split:
        s = Split()
        self.check(self.fn("split_domain")(C.byref(d), C.byref(s)), "split_domain")
        return s

    def radii(self, d: Desc) -> np.ndarray:
        r = np.zeros(d.nr_global + GEOM_PAD + 1)
        self.check(self.fn("radii")(C.byref(d), _as_dp(r)), "radii")
        return r

    def initial_fields(self, d: Desc, radii: np.ndarray):
        """Returns (sigma, vrad, vazi, energy); may update d.sigma0 (SetSigma0)."""
        s = self.split_domain(d)
        sigma = np.zeros((s.nr, d.nphi))
        vrad = np.zeros((s.nr + 1, d.nphi))
        vazi = np.zeros((s.nr, d.nphi))
        energy = np.zeros((s.nr, d.nphi))
        self.check(self.fn("initial_fields")(C.byref(d), _as_dp(radii), _as_dp(sigma), _as_dp(vrad),
                                             _as_dp(vazi), _as_dp(energy)), "initial_fields")
        return sigma, vrad, vazi, energy

    def kernel_names(self):
        n = self.fn("kernel_count")()
        f = self.fn("kernel_name")
        f.restype = C.c_char_p
        return [f(_i32(k)).decode() for k in range(n)]

    def create(self, d: Desc, radii: np.ndarray) -> "Context":
        return Context(self, d, radii)

    def device_count(self) -> int:
        n = _i32()
        self.check(self.fn("device_count")(C.byref(n)), "device_count")
        return n.value

    def set_device(self, device: int):
        self.check(self.fn("set_device")(_i32(device)), "set_device")

    def selftest_half_limiter(self, limiter: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        a, b = (np.ascontiguousarray(v, dtype=np.float64) for v in (a, b))
        out = np.zeros_like(a)
        self.check(self.fn("selftest_half_limiter")(_i32(limiter), C.c_int64(a.size), _as_dp(a), _as_dp(b), _as_dp(out)),
                   "selftest_half_limiter")
        return out

    def selftest_chunk_tables(self, nr: int, nphi: int, n_cu: int = 256, adiabatic: bool = False, damp_inner: int = 0, damp_outer: int = 0):
        """(transport wavefronts [n, 3], source wavefronts [m, 3]) the library would use for such a slab and device: host logic, no GPU."""
        nt, ns = _i32(), _i32()
        f = self.fn("selftest_chunk_tables")
        args = (_i32(nr), _i32(nphi), _i32(n_cu), _i32(int(adiabatic)), _i32(damp_inner), _i32(damp_outer))
        self.check(f(*args, None, _i32(0), C.byref(nt), None, _i32(0), C.byref(ns)), "selftest_chunk_tables")
        t, s = np.zeros((nt.value, 3), dtype=np.int32), np.zeros((ns.value, 3), dtype=np.int32)
        self.check(f(*args, t.ctypes.data_as(C.POINTER(_i32)), _i32(nt.value), C.byref(nt),
                     s.ctypes.data_as(C.POINTER(_i32)), _i32(ns.value), C.byref(ns)), "selftest_chunk_tables")
        return t, s

    def comm_unique_id(self) -> bytes:
        """ncclGetUniqueId: slab 0 calls it and hands the bytes to every slab (fcpt_comm_init)."""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        self.check(self.fn("comm_unique_id")(buf), "comm_unique_id")
        return buf.raw


class Context:
    """Owns one fcpt_ctx (one radial slab on one device)."""

    def __init__(self, lib: Library, d: Desc, radii: np.ndarray):
        self.lib = lib
        self.desc = d.copy()
        self._h = C.c_void_p()
        lib.check(lib.fn("create")(C.byref(self.desc), _as_dp(radii), C.byref(self._h)), "create")
        self.split = Split()
        lib.check(lib.fn("get_split")(self._h, C.byref(self.split)), "get_split")
        self.nr, self.nphi = self.split.nr, d.nphi

    def close(self):
        if self._h:
            self.lib.fn("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        self.lib.check(self.lib.fn(name)(self._h, *args), name)

    def shape(self, field: int):
        return (self.nr + 1, self.nphi) if field in VECTOR_FIELDS else (self.nr, self.nphi)

    def upload(self, field: int, a: np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == self.shape(field), (a.shape, self.shape(field))
        self._call("upload", _i32(field), _as_dp(a))

    def download(self, field: int) -> np.ndarray:
        a = np.zeros(self.shape(field))
        self._call("download", _i32(field), _as_dp(a))
        return a

    def device_ptr(self, field: int):
        p, n = C.c_void_p(), _u64()
        self._call("device_ptr", _i32(field), C.byref(p), C.byref(n))
        return p.value, n.value

    def set_stream(self, stream_handle: int):
        self._call("set_stream", C.c_void_p(stream_handle))

    def synchronize(self):
        self._call("synchronize")

    def set_option(self, name: str, value: int):
        self._call("set_option", C.c_char_p(name.encode()), _i32(value))

    def get_option(self, name: str) -> int:
        v = _i32()
        self._call("get_option", C.c_char_p(name.encode()), C.byref(v))
        return v.value

    def set_transport_chunks(self, lengths):
        """Explicit chunk lengths of the fused transport kernel in dispatch order (empty: the built-in grading)."""
        a = np.ascontiguousarray(lengths, dtype=np.int32)
        self._call("set_transport_chunks", a.ctypes.data_as(C.POINTER(_i32)), _i32(a.size))

    def transport_chunks(self) -> np.ndarray:
        """(tile, first ring, one past the last) of every wavefront of the fused transport kernel, dispatch order; empty: equal chunks."""
        n = _i32()
        self._call("transport_chunks", None, _i32(0), C.byref(n))
        out = np.zeros((n.value, 3), dtype=np.int32)
        if n.value:
            self._call("transport_chunks", out.ctypes.data_as(C.POINTER(_i32)), _i32(n.value), C.byref(n))
        return out

    def source_chunks(self) -> np.ndarray:
        """(segment, first ring, one past the last) of every wavefront of the marching source kernel, dispatch order; empty: equal chunks."""
        n = _i32()
        self._call("source_chunks", None, _i32(0), C.byref(n))
        out = np.zeros((n.value, 3), dtype=np.int32)
        if n.value:
            self._call("source_chunks", out.ctypes.data_as(C.POINTER(_i32)), _i32(n.value), C.byref(n))
        return out

    # radial slabs over RCCL inside the library
    def comm_init(self, unique_id: bytes):
        assert len(unique_id) == COMM_ID_BYTES
        self._call("comm_init", C.c_char_p(unique_id))

    def comm_init_host(self, path: str):
        """The host-staged transport (several slabs on one GPU): a shared-memory file instead of RCCL."""
        self._call("comm_init_host", C.c_char_p(path.encode()))

    def comm_barrier(self):
        self._call("comm_barrier")

    def comm_destroy(self):
        self._call("comm_destroy")

    def exchange(self):
        self._call("exchange")

    def cfl_allreduce(self, blocking: bool = True):
        v = _f64()
        self._call("cfl_allreduce", C.byref(v) if blocking else None)
        return v.value if blocking else None

    def set_bodies(self, x, y, m, rsm=None, indirect=(0.0, 0.0)):
        x, y, m = (np.ascontiguousarray(v, dtype=np.float64) for v in (x, y, m))
        rsm = np.zeros_like(x) if rsm is None else np.ascontiguousarray(rsm, dtype=np.float64)
        self._call("set_bodies", _i32(len(x)), _as_dp(x), _as_dp(y), _as_dp(m), _as_dp(rsm),
                   _f64(indirect[0]), _f64(indirect[1]))

    def set_bodies_midstep(self, x, y, m, rsm=None):
        x, y, m = (np.ascontiguousarray(v, dtype=np.float64) for v in (x, y, m))
        rsm = np.zeros_like(x) if rsm is None else np.ascontiguousarray(rsm, dtype=np.float64)
        self._call("set_bodies_midstep", _i32(len(x)), _as_dp(x), _as_dp(y), _as_dp(m), _as_dp(rsm))

    def init_physics(self):
        self._call("init_physics")

    def set_body_irradiation(self, temperature, radius, rampup_time=None):
        t = np.ascontiguousarray(temperature, dtype=np.float64)
        r = np.ascontiguousarray(radius, dtype=np.float64)
        u = np.ascontiguousarray(rampup_time, dtype=np.float64) if rampup_time is not None else None
        self._call("set_body_irradiation", _i32(len(t)), _as_dp(t), _as_dp(r), _as_dp(u))

    def disk_on_body_accel(self, x, y, r_object, smoothing_fixed=-1.0, cubic_smoothing_radius=0.0):
        """ComputeDiskOnPlanetAccel without the all-reduce: [inner a_x, inner a_y, outer a_x, outer a_y]."""
        out = (C.c_double * 4)()
        self._call("disk_on_body_accel", _f64(x), _f64(y), _f64(r_object), _f64(smoothing_fixed),
                   _f64(cubic_smoothing_radius), out)
        return np.array(out[:], dtype=np.float64)

    def cfl(self) -> float:
        v = _f64()
        self._call("cfl", C.byref(v))
        return v.value

    # device-resident dt (no host synchronisation); arguments are device addresses
    def cfl_device(self, d_dt_local: int):
        self._call("cfl_device", C.c_void_p(int(d_dt_local)))

    def step_device_begin(self):
        self._call("step_device_begin")

    def step_device_end(self):
        self._call("step_device_end")

    def cfl_begin(self):
        self._call("cfl_begin")

    def calculate_timestep_device(self, d_cfl_global: Optional[int] = None):
        self._call("calculate_timestep_device", C.c_void_p(int(d_cfl_global)) if d_cfl_global else None)

    def step_device(self):
        self._call("step_device")

    def post_device(self):
        self._call("post_device")

    def calculate_timestep(self, cfl_dt_global: float) -> float:
        v = _f64()
        self._call("calculate_timestep", _f64(cfl_dt_global), C.byref(v))
        return v.value

    def snap_to_monitor(self, cfl_dt: float) -> float:
        v = _f64()
        self._call("snap_to_monitor", _f64(cfl_dt), C.byref(v))
        return v.value

    def step(self, dt: float):
        self._call("step", _f64(dt))

    def post(self, dt: float):
        self._call("post", _f64(dt))

    def recalculate_derived(self):
        self._call("recalculate_derived")

    def apply_boundary(self, dt: float, final: bool):
        self._call("apply_boundary", _f64(dt), _i32(1 if final else 0))

    def exchange_count(self) -> int:
        n = _u64()
        self._call("exchange_count", C.byref(n))
        return n.value

    def exchange_pack(self, send_inner, send_outer):
        """Arguments: numpy arrays (host libraries) or integer device addresses, or None."""
        self._call("exchange_pack", self._buf(send_inner), self._buf(send_outer))

    def exchange_unpack(self, recv_inner, recv_outer):
        self._call("exchange_unpack", self._buf(recv_inner), self._buf(recv_outer))

    @staticmethod
    def _buf(b):
        if b is None:
            return None
        if isinstance(b, np.ndarray):
            return _as_dp(b)
        return C.cast(C.c_void_p(int(b)), _dp)

    def profile_start(self, kernel_ids=None, max_launches: int = 4096):
        mask = (1 << 64) - 1 if kernel_ids is None else sum(1 << k for k in kernel_ids)
        self._call("profile_start", _u64(mask), _i32(max_launches))

    def profile_stop(self):
        """-> {kernel name: (total ms, launches)} for the kernels that ran."""
        names = self.lib.kernel_names()
        ms = (_f64 * len(names))()
        cnt = (C.c_int64 * len(names))()
        self._call("profile_stop", ms, cnt)
        return {n: (ms[k], cnt[k]) for k, n in enumerate(names) if cnt[k] > 0}

    def run_steps(self, nsteps: int, snap: bool = False) -> int:
        done = C.c_int64()
        self._call("run_steps", C.c_int64(nsteps), _i32(1 if snap else 0), C.byref(done))
        return done.value

    @property
    def clock(self) -> Clock:
        c = Clock()
        self._call("get_clock", C.byref(c))
        return c

    @clock.setter
    def clock(self, c: Clock):
        self._call("set_clock", C.byref(c))

    def dt_statistics(self, reset: bool = False):
        lo, hi = _f64(), _f64()
        self._call("dt_statistics", C.byref(lo), C.byref(hi), _i32(1 if reset else 0))
        return lo.value, hi.value

    def state(self):
        """Host copies of the evolved grids."""
        out = {"sigma": self.download(F_SIGMA), "vrad": self.download(F_VRAD),
               "vazi": self.download(F_VAZI)}
        if self.desc.eos == EOS_IDEAL:
            out["energy"] = self.download(F_ENERGY)
        return out
