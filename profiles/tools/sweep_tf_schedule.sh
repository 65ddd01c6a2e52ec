#!/bin/bash
# ms per step and kernel times of the headline workload for explicit chunk-length lists of k_transport_fused
# (FCPT_TF_SCHEDULE, see transport_schedule() in kernels/launch.h); "uniform" = equal chunks, "auto" = the built-in grading.
# usage: profiles/tools/sweep_tf_schedule.sh "<spec> <spec> ..." [extra bench args]   (run on the GPU box)
SPECS=$1; shift
for v in uniform auto $SPECS; do
  unset FCPT_TF_SCHEDULE FCPT_TRANSPORT_GRADED
  if [ $v = uniform ]; then export FCPT_TRANSPORT_GRADED=0; elif [ $v != auto ]; then export FCPT_TF_SCHEDULE=$v; fi
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-configs --settle-blocks 3 "$@" 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(min(d['ms_per_step_blocks']),4), {k:round(v*1e3,1) for k,v in list(d['kernel_ms_per_step'].items())[:3]})"
done
