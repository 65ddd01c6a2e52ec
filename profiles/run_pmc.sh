#!/bin/bash
# Collects PMC counters for the bench workload in separate rocprofv3 passes
# (kernel-trace only; no sys/hip/hsa tracing together with --pmc).
# usage: profiles/run_pmc.sh <outdir> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --no-cpu-baseline --no-configs --settle-blocks 0 "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed: $set"
done
ls -R $OUT | head -40
