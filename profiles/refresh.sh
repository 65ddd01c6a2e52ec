#!/bin/bash
# Regenerates the judged profile artefacts on the GPU box into gpurun_out/refresh (copy them to profiles/ afterwards):
# rocprofv3 kernel stats of the bench command (isothermal + ideal EOS), the PMC passes, the bench JSON lines.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[1] kernel stats (isothermal)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_iso -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline > $O/stats_iso.json 2> $O/stats_iso.err || echo "stats iso failed"
echo "[2] kernel stats (ideal EOS)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ideal -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --eos ideal > $O/stats_ideal.json 2> $O/stats_ideal.err || echo "stats ideal failed"
find $O -name "*kernel_trace.csv" -delete
echo "[3] PMC passes"
bash $R/profiles/run_pmc.sh $O/pmc --steps 8 --warmup 2 > $O/pmc.log 2>&1
find $O/pmc -name "*kernel_trace.csv" -delete
cd $R
python3 profiles/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
echo "[4] bench lines"
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err || echo "bench failed"
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --eos ideal --no-cpu-baseline > $O/bench_ideal.json 2> $O/bench_ideal.err || echo "bench ideal failed"
ls -la $O | head -30
