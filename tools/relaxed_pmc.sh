#!/bin/bash
# SQ instruction mix of the relaxed kernel: bash tools/relaxed_pmc.sh CHANNELS SECONDS  (through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/relaxed_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -- python3 $R/tools/relaxed_probe.py prof $1 $2 > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python3 $R/tools/relaxed_probe.py prof $1 $2 > $OUT/sq2.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/sq1 $OUT/sq2 > $OUT/summary_$1.txt 2>&1
cat $OUT/summary_$1.txt
