#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter> <python script> [args...] -- one rocprofv3 --pmc pass (no tracing domains), CSV into gpurun_out/<tag>/
set -e
tag=$1; counter=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$tag
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $counter --output-format csv -d $R/gpurun_out/$tag/pmc_$counter -- python3 $script "$@" > $R/gpurun_out/$tag/pmc_$counter.log 2>&1
cd $R
f=$(find gpurun_out/$tag/pmc_$counter -name "*counter_collection.csv" | head -1)
cp $f gpurun_out/$tag/pmc_$counter.csv
echo "$counter -> gpurun_out/$tag/pmc_$counter.csv ($(wc -l < gpurun_out/$tag/pmc_$counter.csv) rows)"
