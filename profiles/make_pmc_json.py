#!/usr/bin/env python3
"""profiles/pmc_latest.json from the rocprofv3 --pmc passes of profiles/run_pmc.sh.
usage: make_pmc_json.py <pmc dir> <summary txt written by pmc_summary.py> [workload]

Per kernel (mean per dispatch): HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE doubled
per MI355X_MICROARCH.md: gfx950 reports half of a coalesced stream), the busy fraction of the
vector ALUs = SQ_ACTIVE_INST_VALU / (1024 SIMDs * kernel quad-cycles), and the mean number of
resident waves per SIMD = SQ_WAVE_CYCLES / (1024 * kernel quad-cycles)."""
import collections, csv, glob, json, re, sys

d = sys.argv[1]
src = sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "2048x4096"
eos = sys.argv[4] if len(sys.argv) > 4 else "isothermal"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/pass*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"workload": workload, "eos": eos, "source": src,
       "note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes "
               "(profiles/run_pmc.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a "
               "coalesced stream; calibrated on k_velocities in round 1). valu_busy = SQ_ACTIVE_INST_VALU / "
               "(1024 SIMDs x GRBM_GUI_ACTIVE/8/4 quad-cycles); waves_per_simd likewise from SQ_WAVE_CYCLES.",
       "hbm_bytes_per_launch": {}, "valu_busy": {}, "waves_per_simd": {}, "valu_insts_per_launch": {}}
mean = lambda v: sum(v) / len(v)
for k, c in acc.items():
    m = re.search(r"(k_[a-z0-9_]+)", k)
    if not m or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    name = m.group(1)
    if mean(c["FETCH_SIZE"]) + mean(c["WRITE_SIZE"]) < 1000:  # tiny kernels
        continue
    out["hbm_bytes_per_launch"][name] = int((2 * mean(c["FETCH_SIZE"]) + mean(c["WRITE_SIZE"])) * 1024)
    if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
        quad = mean(c["GRBM_GUI_ACTIVE"]) / 8 / 4 * 1024
        out["valu_busy"][name] = round(mean(c["SQ_ACTIVE_INST_VALU"]) / quad, 3)
        out["waves_per_simd"][name] = round(mean(c["SQ_WAVE_CYCLES"]) / quad, 2)
        out["valu_insts_per_launch"][name] = int(mean(c["SQ_INSTS_VALU"]))
json.dump(out, sys.stdout, indent=1)
print()
