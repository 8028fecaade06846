#!/bin/bash
# round-6 counter passes over three eager training steps (one rocprofv3 --pmc pass per counter group, no tracing domains)
set -e
bash tools/pmc_multi.sh r06pmc mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" tools/pmc_step.py
bash tools/pmc_multi.sh r06pmc lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" tools/pmc_step.py
bash tools/pmc_multi.sh r06pmc tcc "TCC_HIT_sum TCC_MISS_sum" tools/pmc_step.py
bash tools/pmc_multi.sh r06pmc fetch "FETCH_SIZE" tools/pmc_step.py
bash tools/pmc_multi.sh r06pmc write "WRITE_SIZE" tools/pmc_step.py
python3 tools/pmc_family_summary.py gpurun_out/r06pmc/summary.json gpurun_out/r06pmc/pmc_mfma.csv gpurun_out/r06pmc/pmc_lds.csv gpurun_out/r06pmc/pmc_tcc.csv gpurun_out/r06pmc/pmc_fetch.csv gpurun_out/r06pmc/pmc_write.csv | head -60
python3 tools/fingerprint.py gpurun_out/r06pmc/summary.json
