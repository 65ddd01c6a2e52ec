#!/usr/bin/env python3
"""Parity margins of the library in FCPT_LIB_PATH (or the in-tree one) against the oracle on the cases that sit closest to
the 1e-10 bar: the 4x-CFL step at 2048 x 4096 (isothermal / ideal EOS), 1024 x 3072 ideal EOS over 10 steps, the shock tube
4096 x 4 over 60 steps.  Prints max|a-b|/max|b| per field."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "16")
import numpy as np
import torch  # noqa: F401
import fargocpt_amd
from fargocpt_amd import binding as B, setups
from tests.util import rel_err, run_pair
lib = fargocpt_amd.load()
orc = B.Library(ctypes.CDLL(os.path.join(ROOT, "oracle", "libfargo_oracle.so")), "orc_")
def show(name, res, fields):
    (a, _), (b, _) = res
    print(name, {k: float(f"{rel_err(a[k], b[k]):.3e}") for k in fields}, flush=True)
for adi in (False, True):
    d = setups.planet_disk(lib, 2048, 4096, adiabatic=adi)
    d.damping = 0
    d.first_dt = 1.0
    f = ("sigma", "vrad", "vazi") + (("energy",) if adi else ())
    show(f"4x CFL 2048x4096 {'ideal' if adi else 'iso'}", run_pair(lib, orc, d, 3, bodies=setups.jupiter_bodies(d), dt_scale=4.0), f)
d = setups.planet_disk(lib, 1024, 3072, adiabatic=True)
show("1024x3072 ideal, 10 steps", run_pair(lib, orc, d, 10, bodies=setups.jupiter_bodies(d)), ("sigma", "vrad", "vazi", "energy"))
d = setups.shocktube(lib, 4096, 4, "SN")
d.first_dt = 1e-6
show("shocktube 4096x4, 60 steps", run_pair(lib, orc, d, 60, amp=0.0), ("sigma", "vrad", "vazi", "energy"))
