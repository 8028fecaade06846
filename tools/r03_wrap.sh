#!/bin/bash
# end-of-round evidence at HEAD: the GPU test suite, the default bench command under the kernel trace, the bench line itself
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03w
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc; tail -2 $O/pytest.log
python tests/parity_report.py $O/parity_errors.json > $O/parity.txt 2>&1; tail -1 $O/parity.txt
bash tools/prof_cmd.sh r03w/bench bench.py --skip-cpu > $O/bench_prof.txt 2>&1; echo "bench prof rc=$?"
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
