#!/bin/bash
# round-6 evidence pass, part A on the FINAL binary: step / mim_19 / search kernel-stat profiles, ordered step trace (the full GPU suite ran
# in tools/r06_final_a.sh; here the tests touched since: predictor, f16, kernels)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06f
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_predictor_gpu.py tests/test_f16_gpu.py tests/test_kernels_gpu.py -m gpu -x -q > $O/pytest2.log 2>&1; echo "pytest rc=$?" | tee $O/pytest2.rc; tail -2 $O/pytest2.log
python3 tests/parity_report.py $O/parity_errors.json > /dev/null 2>&1; echo "parity rc=$?"
bash tools/r4_seq.sh r06f/step > $O/step.txt 2>&1; cat $O/step.txt | head -4
SKYEMB_BENCH_NO_AB=1 bash tools/prof_cmd.sh r06f/mim19 tools/mim19_bench.py > $O/mim19.txt 2>&1; tail -3 $O/mim19.txt | cut -c1-200
bash tools/prof_cmd.sh r06f/search tools/search_bench.py 10000 3 > $O/search.txt 2>&1; grep "Q=" $O/search.txt
bash tools/prof_cmd.sh r06f/q16 tools/search_small.py > $O/q16.txt 2>&1; echo "q16 rc=$?"
rm -rf $O/mim19/prof $O/search/prof $O/q16/prof
