#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <python script> [args...]  -- rocprofv3 kernel stats into gpurun_out/<tag>/kernel_stats.csv
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$tag
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $script "$@" > $R/gpurun_out/$tag/prof.log 2>&1
cd $R
f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/$tag/kernel_stats.csv
tail -3 gpurun_out/$tag/prof.log
python3 tools/kstats.py gpurun_out/$tag/kernel_stats.csv 1 14
