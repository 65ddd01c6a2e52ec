#!/bin/bash
# config 3 (1024x3072 ideal EOS): equal chunks of other lengths than the built-in choice, transport and source march
# usage: profiles/tools/sweep_rows_config3.sh   (run on the GPU box)
A="--steps 100 --warmup 20 --no-cpu-baseline --no-configs --settle-blocks 3 --eos ideal --nr 1024 --nphi 3072"
run() { python bench.py $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(min(d['ms_per_step_blocks']),4), {k:round(v*1e3,1) for k,v in list(d['kernel_ms_per_step'].items())[:3]})"; }
run "built-in"
for r in 16 18 20 22 24 28 32; do FCPT_TRANSPORT_ROWS=$r run "transport_rows=$r"; done
for r in 12 14 16 18 20 24 28; do FCPT_SOURCE_ROWS=$r run "source_rows=$r"; done
run "built-in"
