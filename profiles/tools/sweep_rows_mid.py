#!/usr/bin/env python3
"""ms per step against the marching-chunk lengths on grids that do not fill the GPU (calibrates march_rows)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
import fargocpt_amd
from fargocpt_amd import driver, setups
lib = fargocpt_amd.load()
for nr, nphi, adi in ((512, 1536, False), (1024, 3072, True), (128, 384, False), (1024, 2048, False)):
    d = setups.planet_disk(lib, nr, nphi, adiabatic=adi)
    out = []
    for tr, sr in ((-1, -1), (4, 4), (6, 6), (8, 8), (10, 10), (12, 12), (16, 16), (24, 24)):
        ctx = driver.make_context(lib, d, bodies=setups.jupiter_bodies(d))
        ctx.set_option("transport_rows", tr)
        ctx.set_option("source_rows", sr)
        for _ in range(2):
            ctx.calculate_timestep(ctx.cfl())
        ctx.run_steps(30)
        ctx.synchronize()
        t0 = time.perf_counter(); ctx.run_steps(300); ctx.synchronize()
        out.append((tr, round((time.perf_counter() - t0) / 300 * 1e3, 4)))
        ctx.close()
    print(f"{nr}x{nphi} {'ideal' if adi else 'iso'}:", out)
