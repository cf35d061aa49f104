#!/bin/bash
# usage (on the GPU box): tools/attn_pmc.sh [kbench filter]  - SQ counters of the attention kernels (two rocprofv3 --pmc passes of tools/kbench.py attn)
R=$PWD; export PYTHONPATH=$R
F=${1:-attn}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/attn_pmc1 -- python $R/tools/kbench.py "$F" > /dev/null 2>&1
python $R/tools/pmc_sq.py $(find /tmp/attn_pmc1 -name "*counter_collection.csv" | head -1) attn
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d /tmp/attn_pmc2 -- python $R/tools/kbench.py "$F" > /dev/null 2>&1
python $R/tools/pmc_sq.py $(find /tmp/attn_pmc2 -name "*counter_collection.csv" | head -1) attn
