#!/bin/bash
# Issue/stall counters of the dominant kernels for several builds of the library (see ab_multi.sh), separate
# rocprofv3 --pmc passes (kernel trace only).
# usage: profiles/tools/pmc_ab.sh "<name> ..." <outdir> [kernel substring] [extra bench args]   (run on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
NAMES=$1; OUT=$2; KSUB=${3:-k_transport_fused}; shift; shift; shift
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $NAMES; do
  if [ $v = base ]; then unset FCPT_LIB_PATH; else export FCPT_LIB_PATH=$R/fargocpt_amd/libfargocpt_hip_$v.so; fi
  i=0; mkdir -p $OUT/$v
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$v/pass$i -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --settle-blocks 0 "$@" > $OUT/$v/pass$i.log 2>&1 || echo "pass $i of $v failed"
  done
  echo "#### $v"
  python3 $R/profiles/pmc_summary.py $OUT/$v $KSUB
  rm -rf $OUT/$v/pass*/*/*_kernel_trace.csv
done
