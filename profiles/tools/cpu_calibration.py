#!/usr/bin/env python3
"""Cross-calibration of bench.py's cpu_baseline ("kind": "port" = oracle/fargo_oracle.c) against the reference
binary's own timings of BASELINE.md section 2, on the same host (the 8-vCPU Xeon @ 2.10 GHz container the survey
timed the reference in) and the same configuration (examples/config.yml physics, 512 x 1536, isothermal):
reference 3.54 M cell-updates/s with MPI=1/OMP=1 and 11.6 M with OMP=8.  Writes profiles/cpu_calibration.json,
which bench.py attaches to cpu_baseline as "vs_reference"."""
import ctypes, json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = {1: 3.54e6, 8: 11.6e6}   # BASELINE.md section 2, 512 x 1536 isothermal, 7 steps

if len(sys.argv) > 1:   # child: one measurement with the thread count of the environment
    sys.path.insert(0, ROOT)
    import fargocpt_amd
    from fargocpt_amd import binding as B, driver, setups
    lib = fargocpt_amd.load()
    orc = B.Library(ctypes.CDLL(os.path.join(ROOT, "oracle", "libfargo_oracle.so")), "orc_")
    d = setups.planet_disk(lib, 512, 1536)
    ctx = driver.make_context(orc, d, bodies=setups.jupiter_bodies(d))
    for _ in range(2):
        ctx.calculate_timestep(ctx.cfl())
    ctx.run_steps(2)
    n = int(sys.argv[1])
    t0 = time.perf_counter()
    ctx.run_steps(n)
    el = time.perf_counter() - t0
    print(json.dumps({"steps": n, "ms_per_step": 1e3 * el / n, "cell_updates_per_s": 512 * 1536 * n / el}))
    sys.exit(0)

out = {"config": "examples/config.yml physics, 512x1536, isothermal (BASELINE.md section 2)",
       "host": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") + f", {os.cpu_count()} vCPU",
       "reference_cell_updates_per_s": {str(k): v for k, v in REF.items()}, "port_cell_updates_per_s": {}, "port_over_reference": {}}
for threads, steps in ((1, 7), (8, 21)):
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), GOMP_SPINCOUNT="100000")
    best = max(json.loads(subprocess.check_output([sys.executable, __file__, str(steps)], env=env))["cell_updates_per_s"] for _ in range(3))
    out["port_cell_updates_per_s"][str(threads)] = best
    out["port_over_reference"][str(threads)] = best / REF[threads]
out["note"] = ("the oracle is compiled -O2 -ffp-contract=off (a plain IEEE restatement), the reference -Ofast -march=native -flto; "
               "a cpu_baseline value of the port divided by port_over_reference estimates what the reference binary would do on the same cores")
json.dump(out, open(os.path.join(ROOT, "profiles", "cpu_calibration.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
