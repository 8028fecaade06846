#!/bin/bash
# round-6 counter passes over the two other workloads: the many-query search (Q = 10 000) and two eager mim_19 steps
# (one rocprofv3 --pmc pass per counter group, no tracing domains)
set -e
for tgt in "search tools/search_bench.py 10000 1" "mim19 tools/mim19_pmc.py"; do
    set -- $tgt; name=$1; shift
    bash tools/pmc_multi.sh r06x_$name mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "$@"
    bash tools/pmc_multi.sh r06x_$name tcc "TCC_HIT_sum TCC_MISS_sum" "$@"
    bash tools/pmc_multi.sh r06x_$name fetch "FETCH_SIZE" "$@"
    bash tools/pmc_multi.sh r06x_$name write "WRITE_SIZE" "$@"
    python3 tools/pmc_kernel_summary.py gpurun_out/r06x_$name/summary.json "rocprofv3 --pmc <counters> --output-format csv -- python3 $* (tools/r06_pmc_extra.sh: one pass per counter group, no tracing domains), summarised by tools/pmc_kernel_summary.py" 0.01 gpurun_out/r06x_$name/pmc_mfma.csv gpurun_out/r06x_$name/pmc_tcc.csv gpurun_out/r06x_$name/pmc_fetch.csv gpurun_out/r06x_$name/pmc_write.csv > gpurun_out/r06x_$name/summary.txt
    python3 tools/fingerprint.py gpurun_out/r06x_$name/summary.json
done
