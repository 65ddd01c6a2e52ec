#!/bin/bash
# A/B of two builds of the library on the bench workload: alternating runs, ms per step of the settled blocks
# usage: profiles/tools/ab_lib.sh <alt .so> [pairs] [extra bench args]   (run on the GPU box)
ALT=$1; N=${2:-3}; shift; shift
for i in $(seq 1 $N); do
  for v in base alt; do
    if [ $v = alt ]; then export FCPT_LIB_PATH=$ALT; else unset FCPT_LIB_PATH; fi
    python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-configs --settle-blocks 3 "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(min(d['ms_per_step_blocks']),4), d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), {k:round(v*1e3,1) for k,v in list(d['kernel_ms_per_step'].items())[:4]})"
  done
done
