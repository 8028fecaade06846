#!/bin/bash
# usage: tools/prof_step.sh <tag> [env assignments...]   -- rocprofv3 kernel stats of tools/step_only.py into gpurun_out/<tag>/
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$tag
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/tools/step_only.py > $R/gpurun_out/$tag/prof.log 2>&1
cd $R
f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/$tag/kernel_stats.csv
grep "ms/step" gpurun_out/$tag/prof.log
python3 tools/kstats.py gpurun_out/$tag/kernel_stats.csv 33 18
