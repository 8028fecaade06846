#!/bin/bash
# round-6 evidence pass, part C: the whole GPU suite on the final binary (parity record) and the bench line with the round-6 counter files in place
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06f
mkdir -p $O
cd $R
rm -f gpurun_out/parity_errors.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_final.log 2>&1; echo "pytest rc=$?" | tee $O/pytest_final.rc; tail -2 $O/pytest_final.log
python3 tests/parity_report.py $O/parity_errors_final.json > /dev/null 2>&1; echo "parity rc=$?"
python3 __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python3 bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; tail -c 200 $O/bench_final.err
