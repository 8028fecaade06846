#!/bin/bash
set -e
bash tools/pmc_multi.sh r03pmc mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" tools/pmc_step.py
bash tools/pmc_multi.sh r03pmc lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" tools/pmc_step.py
bash tools/pmc_multi.sh r03pmc tcc "TCC_HIT_sum TCC_MISS_sum" tools/pmc_step.py
bash tools/pmc_multi.sh r03pmc fetch "FETCH_SIZE" tools/pmc_step.py
bash tools/pmc_multi.sh r03pmc write "WRITE_SIZE" tools/pmc_step.py
python3 tools/pmc_family_summary.py gpurun_out/r03pmc/summary.json gpurun_out/r03pmc/pmc_mfma.csv gpurun_out/r03pmc/pmc_lds.csv gpurun_out/r03pmc/pmc_tcc.csv gpurun_out/r03pmc/pmc_fetch.csv gpurun_out/r03pmc/pmc_write.csv | head -120
