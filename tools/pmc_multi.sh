#!/bin/bash
# usage: tools/pmc_multi.sh <tag> <name> "<counters...>" <python script> [args...] -- one rocprofv3 --pmc pass (several counters of
# one pass, no tracing domains; program directly after --), CSV into gpurun_out/<tag>/pmc_<name>.csv
set -e
tag=$1; name=$2; counters=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$tag
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $counters --output-format csv -d $R/gpurun_out/$tag/pmc_$name -- python3 $script "$@" > $R/gpurun_out/$tag/pmc_$name.log 2>&1
cd $R
f=$(find gpurun_out/$tag/pmc_$name -name "*counter_collection.csv" | head -1)
cp $f gpurun_out/$tag/pmc_$name.csv
rm -rf gpurun_out/$tag/pmc_$name
echo "$name ($counters) -> gpurun_out/$tag/pmc_$name.csv ($(wc -l < gpurun_out/$tag/pmc_$name.csv) rows)"
