#!/bin/bash
# ordered per-launch trace of the step: gpurun_out/r4/seq.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r4/seq}
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/tools/step_only.py > $R/gpurun_out/$tag/prof.log 2>&1
cd $R
f=$(find gpurun_out/$tag/prof -name "*kernel_trace.csv" | head -1)
python3 tools/step_sequence.py $f > gpurun_out/$tag/seq.txt
python3 tools/trace_gaps.py $f 60 > gpurun_out/$tag/gaps.txt
cp $(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1) gpurun_out/$tag/kernel_stats.csv
rm -rf gpurun_out/$tag/prof
grep "ms/step" gpurun_out/$tag/prof.log
head -3 gpurun_out/$tag/gaps.txt
