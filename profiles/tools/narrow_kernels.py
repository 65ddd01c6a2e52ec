#!/usr/bin/env python3
"""Per-kernel HIP-event times and launch counts of one step on the launch-bound BASELINE grids (config 5: shock tube
4096x4, config 1: spreading ring 128x384) -- which launches a step of a grid that cannot fill the GPU is made of."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (the HIP runtime torch bundles first)
import fargocpt_amd
from fargocpt_amd import driver, setups
lib = fargocpt_amd.load()
cases = [("5: shocktube 4096x4 SN", setups.shocktube(lib, 4096, 4, "SN"), None),
         ("1: spreading ring 128x384", setups.spreading_ring(lib, 128, 384), None),
         ("shocktube 100x2 SN (the reference's test grid)", setups.shocktube(lib, 100, 2, "SN"), None)]
for name, d, bodies in cases:
    ctx = driver.make_context(lib, d, bodies=bodies)
    for _ in range(2):
        ctx.calculate_timestep(ctx.cfl())
    ctx.run_steps(300)
    ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_steps(2000); ctx.synchronize(); ms = (time.perf_counter() - t0) / 2000 * 1e3
    ctx.profile_start(None, max_launches=800)
    ctx.run_steps(10)
    p = ctx.profile_stop()
    n = sum(v[1] for v in p.values()) / 10
    print(f"{name}: {ms * 1e3:.1f} us/step (graph replays {ctx.get_option('graph_replays')}), {n:.0f} launches per step;",
          {k: (round(v[0] * 100, 1), v[1] // 10) for k, v in sorted(p.items(), key=lambda kv: -kv[1][0])})
    ctx.close()
