#!/bin/bash
# alternating A/B of chunk schedules of k_transport_fused: "uniform", "auto" or an explicit FCPT_TF_SCHEDULE list
# usage: profiles/tools/ab_tf_schedule.sh "<variant> ..." rounds [extra bench args]   (run on the GPU box)
VARS=$1; N=$2; shift; shift
for i in $(seq 1 $N); do
  for v in $VARS; do
    unset FCPT_TF_SCHEDULE FCPT_TRANSPORT_GRADED
    if [ $v = uniform ]; then export FCPT_TRANSPORT_GRADED=0; elif [ $v != auto ]; then export FCPT_TF_SCHEDULE=$v; fi
    python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-configs --settle-blocks 3 "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(min(d['ms_per_step_blocks']),4), {k:round(v*1e3,1) for k,v in list(d['kernel_ms_per_step'].items())[:3]})"
  done
done
