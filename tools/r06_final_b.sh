#!/bin/bash
# round-6 evidence pass, part B: counter passes (step, mim_19, many-query search, Q = 16 bank pass), the bench command under the profiler and plain
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06f
mkdir -p $O
cd $R
timeout -k 10 500 bash tools/r06_pmc.sh > $O/pmc.txt 2>&1; echo "pmc rc=$?"
cp gpurun_out/r06pmc/summary.json $O/mfma_pmc_summary.json 2>/dev/null
rm -f gpurun_out/r06pmc/pmc_*.csv          # (raw per-dispatch counter rows: tens of MB; gpurun brings back at most 64 MiB)
timeout -k 10 600 bash tools/r06_pmc_extra.sh > $O/pmc_extra.txt 2>&1; echo "pmc extra rc=$?"
cp gpurun_out/r06x_search/summary.json $O/search_pmc_summary.json 2>/dev/null; cp gpurun_out/r06x_mim19/summary.json $O/mim19_pmc_summary.json 2>/dev/null
rm -f gpurun_out/r06x_search/pmc_*.csv gpurun_out/r06x_mim19/pmc_*.csv
timeout -k 10 200 bash tools/r06_search_pmc.sh > $O/pmc_q16.txt 2>&1; echo "pmc q16 rc=$?"
cp gpurun_out/r06spmc/r06_topk_stream_pmc.json $O/ 2>/dev/null
rm -f gpurun_out/r06spmc/pmc_*.csv
bash tools/prof_cmd.sh r06f/bench bench.py --skip-cpu > $O/bench_prof.txt 2>&1; echo "bench prof rc=$?"
rm -rf $O/bench/prof
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
du -sh gpurun_out | tail -1
