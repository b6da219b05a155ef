#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run from the repo root through gpurun):
#   bash tools/profile_round.sh            -> gpurun_out/prof/{trace,fetch,write}/..., gpurun_out/prof/*.log
# Kernel trace + stats and the two PMC passes are separate rocprofv3 runs of bench.py itself (the program
# directly after `--`); every run covers all of the bench line's blocks: configs[1] in strict and in
# time-parallel / relaxed modes, the 32 768-channel shard of configs[3] (strict and relaxed), configs[2] at 48 kHz.
# Every command under its own timeout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --check 0 --no-carried-state-check > $OUT/trace_bench.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --no-cpu-baseline --check 0 --no-carried-state-check --preheat-ms 0 --steps 2 --warmup 1 > $OUT/fetch_bench.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --no-cpu-baseline --check 0 --no-carried-state-check --preheat-ms 0 --steps 2 --warmup 1 > $OUT/write_bench.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python3 $R/bench.py --no-cpu-baseline --check 0 --no-carried-state-check --preheat-ms 0 --steps 2 --warmup 1 > $OUT/sq_bench.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/sq > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*.csv" | head -40 > $OUT/files.txt
du -sh $OUT >> $OUT/files.txt
