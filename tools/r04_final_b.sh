#!/bin/bash
# round-4 evidence pass, part B: counter passes, the bench command under the profiler and plain
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout -k 10 500 bash tools/r04_pmc.sh > $O/pmc.txt 2>&1; echo "pmc rc=$?"
cp gpurun_out/r04pmc/summary.json $O/mfma_pmc_summary.json 2>/dev/null
bash tools/prof_cmd.sh r04f/bench bench.py --skip-cpu > $O/bench_prof.txt 2>&1; echo "bench prof rc=$?"
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 tools/search_pieces.py > $O/search_pieces.txt 2>&1; tail -12 $O/search_pieces.txt
