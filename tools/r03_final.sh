#!/bin/bash
# round-3 evidence pass at HEAD: tests, kernel-stat profiles, GEMM timeline, PMC summaries, the bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc; tail -2 $O/pytest.log
bash tools/prof_step.sh r03f/step > $O/step.txt 2>&1; tail -4 $O/step.txt | head -2
bash tools/prof_cmd.sh r03f/mim19 tools/mim19_bench.py > $O/mim19.txt 2>&1
bash tools/prof_cmd.sh r03f/search tools/search_bench.py 10000 3 > $O/search.txt 2>&1
bash tools/prof_cmd.sh r03f/q16 tools/search_small.py > $O/q16.txt 2>&1
LAB_NOCHECK=1 LAB_STAMP=1 timeout -k 10 120 tools/ubench/gemm_lab_stamp 64064 > $O/gemm_timeline.json 2> $O/gemm_timeline.err; echo "stamp rc=$?"
timeout -k 10 400 bash tools/r03_pmc.sh > $O/pmc.txt 2>&1; echo "pmc rc=$?"
cp gpurun_out/r03pmc/summary.json $O/mfma_pmc_summary.json 2>/dev/null
bash tools/prof_cmd.sh r03f/bench bench.py --skip-cpu > $O/bench_prof.txt 2>&1; echo "bench prof rc=$?"
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 tools/search_pieces.py > $O/search_pieces.txt 2>&1; cat $O/search_pieces.txt
