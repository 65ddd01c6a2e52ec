"""Diagnostic: growth of the HIP-vs-oracle difference with StabilizeViscosity (run on the GPU box)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import fargocpt_amd
from fargocpt_amd import binding as B, setups
from fargocpt_amd.binding import Library
from tests.util import run_pair, rel_err

P = fargocpt_amd.load()
O = Library(ctypes.CDLL(os.path.join(ROOT, "oracle", "libfargo_oracle.so")), "orc_")
for mode, scale in ((1, 3.0), (0, 3.0), (1, 1.0)):
    for n in (1, 2, 5, 10, 20, 40, 70):
        d = setups.planet_disk(P, 48, 192)
        d.viscous_alpha, d.constant_viscosity = 0.0, 1.0e-2
        d.stabilize_viscosity = mode
        (a, dta), (b, dtb) = run_pair(P, O, d, n, dt_scale=scale)
        print(mode, scale, n, " ".join(f"{k}:{rel_err(a[k], b[k]):.2e}" for k in ("sigma", "vrad", "vazi")), flush=True)
