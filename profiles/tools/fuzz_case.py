"""Re-run single draws of tests/test_gpu_fuzz.py and print the configuration and the HIP-vs-oracle differences.
usage: FCPT_FUZZ_WIDE=1 python profiles/tools/fuzz_case.py <seed> [<seed> ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import fargocpt_amd
from fargocpt_amd import binding as B, setups
from fargocpt_amd.binding import Library
from tests.util import run_pair, rel_err
from tests.test_gpu_fuzz import draw

P = fargocpt_amd.load()
O = Library(ctypes.CDLL(os.path.join(ROOT, "oracle", "libfargo_oracle.so")), "orc_")
for seed in map(int, sys.argv[1:]):
    d, nslabs, planet = draw(P, seed)
    keys = ("nr_global", "nphi", "radial_spacing", "eos", "viscous_alpha", "constant_viscosity", "artificial_viscosity",
            "integrator", "flux_limiter", "fast_transport", "omega_frame", "cfl", "damping", "stabilize_viscosity",
            "cooling_surface", "cooling_beta", "profile_cutoff_outer", "profile_cutoff_inner")
    print(f"seed {seed}: slabs {nslabs} planet {planet} " + " ".join(f"{k}={getattr(d, k)}" for k in keys))
    print("   bc", list(d.bc_sigma), list(d.bc_energy), list(d.bc_vrad), list(d.bc_vaz), "damp", list(d.damp_vrad), list(d.damp_vaz), list(d.damp_sigma), list(d.damp_energy))
    bodies = setups.jupiter_bodies(d) if planet else None
    for ns in (nslabs,):
        for nsteps in (1, 3, 10):
            (a, dta), (b, dtb) = run_pair(P, O, d, nsteps, bodies=bodies, nslabs=(ns, 1))
            errs = {k: rel_err(a[k], b[k]) for k in ("sigma", "vrad", "vazi", "energy")}
            print(f"   slabs {ns} steps {nsteps}: dt rel diff {max(abs(x - y) / y for x, y in zip(dta, dtb)):.2e} " +
                  " ".join(f"{k}:{v:.2e}" for k, v in errs.items()), flush=True)
