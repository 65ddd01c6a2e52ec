#!/usr/bin/env python3
"""Times fcpt_disk_on_body_accel (ComputeDiskOnPlanetAccel) on the bench grid: mean over 50 calls,
each blocking until its four sums are on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (HIP runtime first)
import fargocpt_amd
from fargocpt_amd import driver, setups

lib = fargocpt_amd.load()
for adi in (False, True):
    d = setups.planet_disk(lib, 2048, 4096, adiabatic=adi)
    ctx = driver.make_context(lib, d, bodies=setups.jupiter_bodies(d))
    ctx.run_steps(3, snap=False)
    ctx.synchronize()
    for mode, sm in (("H per cell", -1.0), ("fixed", 0.03)):
        ctx.disk_on_body_accel(1.0, 0.0, 1.0, sm)
        t0 = time.perf_counter()
        for _ in range(50):
            a = ctx.disk_on_body_accel(1.0, 0.0, 1.0, sm)
        dt = (time.perf_counter() - t0) / 50
        print(f"{'ideal' if adi else 'isothermal'} 2048x4096, smoothing {mode}: {dt*1e6:.1f} us per call, a = {a}")
    ctx.close()
