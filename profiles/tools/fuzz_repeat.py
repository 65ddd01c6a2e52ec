"""Is the HIP result of a fuzz draw reproducible run to run (bitwise)?  usage: FCPT_FUZZ_WIDE=1 fuzz_repeat.py <seed> [n]"""
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
seed = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d, nslabs, planet = draw(P, seed)
bodies = setups.jupiter_bodies(d) if planet else None
ref = run_pair(O, O, d, 10, bodies=bodies, nslabs=(1, 0))[0][0]
first = None
for k in range(n):
    a = run_pair(P, P, d, 10, bodies=bodies, nslabs=(nslabs, 0))[0][0]
    if first is None:
        first = a
    print(k, "vs first:", {f: float(np.abs(a[f] - first[f]).max()) for f in ("sigma", "vrad", "vazi")},
          "vs oracle:", {f: f"{rel_err(a[f], ref[f]):.2e}" for f in ("sigma", "vrad", "vazi")}, flush=True)
