#!/bin/bash
# round-3 first GPU pass: tests, GEMM timeline, staged / overlap pricing, step profile
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
timeout -k 10 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -3 $O/pytest.log
LAB_NOCHECK=1 LAB_STAMP=1 timeout -k 10 120 tools/ubench/gemm_lab_stamp 64064 > $O/gemm_timeline.json 2> $O/gemm_timeline.err; echo "stamp rc=$?"
head -c 600 $O/gemm_timeline.json
for cfg in "mono" "SKYEMB_STAGED=1" "SKYEMB_STAGED=1 SKYEMB_OPT_OVERLAP=1"; do
  echo "== $cfg"
  if [ "$cfg" = "mono" ]; then timeout -k 10 120 python3 tools/step_only.py 2>&1 | tail -1; else env $cfg timeout -k 10 120 python3 tools/step_only.py 2>&1 | tail -1; fi
done
bash tools/prof_step.sh r03a/step 2>&1 | tail -25
