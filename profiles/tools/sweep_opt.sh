#!/bin/bash
# sweep of one kernel-selection option (its FCPT_<NAME> default) on the bench workload
# usage: profiles/tools/sweep_opt.sh NAME "v1 v2 ..." [extra bench args]   (run on the GPU box)
NAME=$1; VALS=$2; shift; shift
for v in $VALS; do
  env FCPT_$NAME=$v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-configs --settle-blocks 3 "$@" 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$NAME=$v', round(min(d['ms_per_step_blocks']),4), {k:round(v*1e3,1) for k,v in list(d['kernel_ms_per_step'].items())[:5]})"
done
