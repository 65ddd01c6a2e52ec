#!/bin/bash
# usage: bench_variants.sh "<label>|<env assignments>" ...   (runs bench.py per variant, prints ms/step and per-kernel ms)
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  env $envs timeout -k 10 150 python bench.py --no-cpu-baseline --steps ${STEPS:-50} --warmup 5 > gpurun_out/bench_$label.json 2>&1
  python - "$label" <<'PY'
import json,sys
lab=sys.argv[1]
ok=False
for l in open(f"gpurun_out/bench_{lab}.json"):
    if l.startswith("{"):
        d=json.loads(l); ok=True
        print(lab, round(d["ms_per_step"],4), "G=%.2f"%(d["value"]/1e9), {k[2:]:round(x,4) for k,x in d["kernel_ms_per_step"].items()})
if not ok: print(lab, "FAILED")
PY
done
