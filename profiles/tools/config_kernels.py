#!/usr/bin/env python3
"""Per-kernel HIP-event times of one step at the BASELINE mid-size configurations (512x1536 isothermal, 1024x3072 ideal):
where a step goes when the grid does not fill the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (the HIP runtime torch bundles first)
import fargocpt_amd
from fargocpt_amd import driver, setups
lib = fargocpt_amd.load()
for nr, nphi, adi in ((512, 1536, False), (1024, 3072, True), (128, 384, False)):
    d = setups.planet_disk(lib, nr, nphi, adiabatic=adi)
    ctx = driver.make_context(lib, d, bodies=setups.jupiter_bodies(d))
    for _ in range(2):
        ctx.calculate_timestep(ctx.cfl())
    ctx.run_steps(30)
    ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_steps(200); ctx.synchronize(); ms = (time.perf_counter() - t0) / 200 * 1e3
    ctx.profile_start(None, max_launches=400)
    ctx.run_steps(10)
    p = ctx.profile_stop()
    print(f"{nr}x{nphi} {'ideal' if adi else 'iso'}: {ms:.4f} ms/step;", {k: (round(v[0] * 100, 1), v[1] // 10) for k, v in sorted(p.items(), key=lambda kv: -kv[1][0])})
    ctx.close()
