#!/bin/bash
# round-5 counter passes over three eager training steps (one rocprofv3 --pmc pass per counter group, no tracing domains)
set -e
bash tools/pmc_multi.sh r05pmc mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" tools/pmc_step.py
bash tools/pmc_multi.sh r05pmc lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" tools/pmc_step.py
bash tools/pmc_multi.sh r05pmc tcc "TCC_HIT_sum TCC_MISS_sum" tools/pmc_step.py
bash tools/pmc_multi.sh r05pmc fetch "FETCH_SIZE" tools/pmc_step.py
bash tools/pmc_multi.sh r05pmc write "WRITE_SIZE" tools/pmc_step.py
python3 tools/pmc_family_summary.py gpurun_out/r05pmc/summary.json gpurun_out/r05pmc/pmc_mfma.csv gpurun_out/r05pmc/pmc_lds.csv gpurun_out/r05pmc/pmc_tcc.csv gpurun_out/r05pmc/pmc_fetch.csv gpurun_out/r05pmc/pmc_write.csv | head -60
python3 tools/fingerprint.py gpurun_out/r05pmc/summary.json
