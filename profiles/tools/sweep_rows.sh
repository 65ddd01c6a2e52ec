#!/bin/bash
# marching-chunk lengths against wave quantisation: ms per step of the bench workload for a list of
# FCPT_TRANSPORT_ROWS / FCPT_SOURCE_ROWS values (run on the GPU box; writes gpurun_out/sweep_rows.txt)
out=gpurun_out/sweep_rows.txt
: > $out
for tr in ${TR_LIST:-24 20 28 32 36 40 44 52}; do
  FCPT_TRANSPORT_ROWS=$tr python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-configs --settle-blocks 2 $EXTRA 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('transport_rows', $tr, 'ms', min(d['ms_per_step_blocks']), 'kernel', d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
done
for sr in ${SR_LIST:-24 16 20 29 32 41}; do
  FCPT_SOURCE_ROWS=$sr python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-configs --settle-blocks 2 $EXTRA 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('source_rows', $sr, 'ms', min(d['ms_per_step_blocks']), 'src_ms', d['kernel_ms_per_step'].get('k_source_march'))" >> $out
done
cat $out
