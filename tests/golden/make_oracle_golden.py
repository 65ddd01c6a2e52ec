#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_reference_runs.json from the CPU oracle.

Runs the reference's known-answer setups (test/shockTube/setups/shocktube_{SN,TW}.yml,
test/spreading_ring/setup.yml, test/steady_state_accretion/setup.yml, test/cold_disk(_planet)/setup.yml) to their snapshot time and records step counts, deviation
metrics and field checksums.  Takes ~2 min."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "4")

import fargocpt_amd  # noqa: E402
from fargocpt_amd import binding as B, driver, setups  # noqa: E402
from tests.known_answers import (run_cold_disk, run_steady_accretion, shocktube_deviations, spreading_ring_deviation,  # noqa: E402
                                 steady_accretion_deviation)


def main():
    lib = fargocpt_amd.load()
    orc = B.Library(ctypes.CDLL(os.path.join(ROOT, "oracle", "libfargo_oracle.so")), "orc_")
    out = {}
    for av, lf in (("SN", False), ("TW", False), ("SN", True), ("TW", True)):
        d = setups.shocktube(lib, 100, 2, av, leapfrog=lf)
        ctx = driver.make_context(orc, d)
        s = driver.SlabSet([ctx])
        s.prepare()
        n = ctx.run_steps(100000, snap=True)
        st = ctx.state()
        out[f"shocktube_{av}" + ("_LF" if lf else "")] = {
            "steps": n, "time": ctx.clock.time, "deviations": shocktube_deviations(lib, d, ctx),
            "sum_sigma": float(st["sigma"].sum()), "sum_energy": float(st["energy"].sum()),
            "max_vrad": float(np.abs(st["vrad"]).max())}
    d = setups.spreading_ring(lib, 256, 2)
    ctx = driver.make_context(orc, d)
    s = driver.SlabSet([ctx])
    s.prepare()
    n = ctx.run_steps(100000, snap=True)
    st = ctx.state()
    out["spreading_ring_256x2"] = {
        "steps": n, "time": ctx.clock.time, "mean_rel_deviation": spreading_ring_deviation(lib, d, ctx),
        "sum_sigma": float(st["sigma"].sum()), "max_vrad": float(np.abs(st["vrad"]).max())}
    d = setups.steady_state_accretion(lib)
    mf, steps = run_steady_accretion(orc, lib, d)
    out["steady_state_accretion_198x1"] = {
        "steps": steps, "max_rel_deviation": steady_accretion_deviation(lib, d, mf),
        "massflow_code_units_at_interface_100": float(mf[100])}
    out["cold_disk"] = run_cold_disk(orc, lib, planet=False)          # 20 orbits, ~10 s
    out["cold_disk_planet"] = run_cold_disk(orc, lib, planet=True)    # 100 orbits, ~1 min
    out["cold_disk_planet_first_snapshot"] = run_cold_disk(orc, lib, planet=True, max_snapshots=1)
    path = os.path.join(ROOT, "tests", "golden", "oracle_reference_runs.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
