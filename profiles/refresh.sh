#!/bin/bash
# Regenerates the judged profile artefacts on the GPU box into gpurun_out/refresh (copy them to profiles/ afterwards):
# rocprofv3 kernel stats of the bench command (isothermal + ideal EOS), the PMC passes, the bench JSON lines.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-configs --settle-blocks 0"
echo "[1] kernel stats (isothermal)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_iso -- python3 $R/bench.py --steps 60 --warmup 10 $B > $O/stats_iso.json 2> $O/stats_iso.err || echo "stats iso failed"
echo "[2] kernel stats (ideal EOS)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ideal -- python3 $R/bench.py --steps 40 --warmup 10 $B --eos ideal > $O/stats_ideal.json 2> $O/stats_ideal.err || echo "stats ideal failed"
find $O -name "*kernel_trace.csv" -delete
echo "[3] PMC passes (isothermal, ideal EOS)"
bash $R/profiles/run_pmc.sh $O/pmc --steps 8 --warmup 2 > $O/pmc.log 2>&1
bash $R/profiles/run_pmc.sh $O/pmc_ideal --steps 8 --warmup 2 --eos ideal > $O/pmc_ideal.log 2>&1
cd $R
python3 profiles/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
python3 profiles/pmc_summary.py $O/pmc_ideal > $O/pmc_ideal_summary.txt 2>&1
python3 profiles/make_pmc_json.py $O/pmc profiles/r03_pmc_summary.txt 2048x4096 isothermal > $O/pmc_latest.json
python3 profiles/make_pmc_json.py $O/pmc_ideal profiles/r03_ideal_eos_pmc_summary.txt 2048x4096 ideal > $O/pmc_latest_ideal.json
find $O/pmc $O/pmc_ideal -name "*kernel_trace.csv" -delete
find $O/pmc $O/pmc_ideal -name "*counter_collection.csv" -delete
echo "[4] bench lines"
timeout -k 10 400 python3 bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err || echo "bench failed"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-configs > $O/bench_driver_form.json 2> $O/bench_driver_form.err || echo "bench (driver form) failed"
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --eos ideal --no-cpu-baseline --no-configs > $O/bench_ideal.json 2> $O/bench_ideal.err || echo "bench ideal failed"
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --rehearse-exchange > $O/bench_rehearse.json 2> $O/bench_rehearse.err || echo "bench rehearse failed"
ls -la $O | head -40
