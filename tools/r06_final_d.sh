#!/bin/bash
# round-6 evidence pass, part D: the whole GPU suite on the final tree -> the parity record
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06f
mkdir -p $O
cd $R
rm -f gpurun_out/parity_errors.json
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_final.log 2>&1; echo "pytest rc=$?" | tee $O/pytest_final.rc; tail -4 $O/pytest_final.log
python3 tests/parity_report.py $O/parity_errors_final.json > /dev/null 2>&1; echo "parity rc=$?"
grep "\[parity\] mim19_full_depth" $O/pytest_final.log
