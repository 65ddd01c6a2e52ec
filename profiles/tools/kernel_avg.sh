#!/bin/bash
# rocprofv3 average duration of selected kernels for the in-tree build and an alt build (FCPT_LIB_PATH): two bench runs each
# usage: profiles/tools/kernel_avg.sh <alt name> "<kernel name fragment> ..." [bench args]   (run on the GPU box)
ALT=$1; KS=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in base $ALT base $ALT; do
  if [ $v = base ]; then unset FCPT_LIB_PATH; else export FCPT_LIB_PATH=$R/fargocpt_amd/libfargocpt_hip_$v.so; fi
  D=$R/gpurun_out/kavg_$v
  rm -rf $D
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --settle-blocks 0 "$@" > /dev/null 2>&1
  for k in $KS; do grep -h "$k" $D/*/*kernel_stats.csv | awk -F, -v v=$v -v k=$k '{printf "%s %s calls %s avg_us %.2f\n", v, k, $(NF-6), $(NF-4)/1000}' | head -2; done
  rm -rf $D
done
