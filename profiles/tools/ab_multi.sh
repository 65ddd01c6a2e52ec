#!/bin/bash
# A/B of several builds of the library (make alt ALTNAME=<name>) against the in-tree one: alternating runs, ms per step
# of the settled blocks and the leading kernels.
# usage: profiles/tools/ab_multi.sh "<name> <name> ..." [rounds] [extra bench args]   (run on the GPU box)
NAMES=$1; N=${2:-2}; shift; shift
for i in $(seq 1 $N); do
  for v in base $NAMES; do
    if [ $v = base ]; then unset FCPT_LIB_PATH; else export FCPT_LIB_PATH=$PWD/fargocpt_amd/libfargocpt_hip_$v.so; fi
    python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-configs --settle-blocks 3 "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(min(d['ms_per_step_blocks']),4), 'parity', d.get('parity_max_rel'), {k:round(v*1e3,1) for k,v in list(d['kernel_ms_per_step'].items())[:4]})"
  done
done
