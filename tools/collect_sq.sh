#!/bin/bash
# Run ON the GPU box (via gpurun): MFMA-pipe utilisation and SQ counters of every kernel of the bench step, three separate --pmc passes
# (counters only: no trace domains in a PMC run).   tools/collect_sq.sh <tag> [extra bench.py arguments]
tag=${1:-x}
shift
extra="$@"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py $extra --steps 3 --warmup 1 --no-profile --no-cpu-baseline --no-extras"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/sq_${tag}_m -- $B > /dev/null 2>&1
python $R/tools/pmc_mfma_util.py $(find /tmp/sq_${tag}_m -name "*counter_collection.csv" | head -1) $R/gpurun_out/${tag}_mfma_util.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/sq_${tag}_1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/sq_${tag}_2 -- $B > /dev/null 2>&1
f1=$(find /tmp/sq_${tag}_1 -name "*counter_collection.csv" | head -1); f2=$(find /tmp/sq_${tag}_2 -name "*counter_collection.csv" | head -1)
{ python $R/tools/pmc_sq.py $f1; python $R/tools/pmc_sq.py $f2; } > $R/gpurun_out/${tag}_sq_counters.txt
python $R/tools/sq_summary.py $f1 $f2 > $R/gpurun_out/${tag}_sq_summary.txt
head -30 $R/gpurun_out/${tag}_sq_summary.txt
