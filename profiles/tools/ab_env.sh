#!/bin/bash
# A/B of one option of the library through its FCPT_<NAME> environment default: alternating runs, ms per step of the
# settled blocks and the leading kernels.
# usage: profiles/tools/ab_env.sh FCPT_INLINE_POTENTIAL "1 0" [rounds] [extra bench args]   (run on the GPU box)
VAR=$1; VALS=$2; N=${3:-2}; shift; shift; shift
for i in $(seq 1 $N); do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-configs --settle-blocks 3 "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(min(d['ms_per_step_blocks']),4), {k:round(v*1e3,1) for k,v in list(d['kernel_ms_per_step'].items())[:4]})"
  done
done
