#!/bin/bash
# Kernel times (HIP events, tools/run_once.py) at the configurations the bench line does not carry, plus a rocprofv3
# kernel trace of the largest one:   bash tools/other_configs.sh  -> gpurun_out/other/configs.txt
# Every run under its own timeout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/other
mkdir -p $OUT
cd $R
{
echo "### channel-count sweep at 22.05 kHz, 2 s per launch, strict (kernel chosen by the dispatcher; dense = one-wavefront kernel built for two wavefronts per SIMD)"
for c in 1024 2048 4096 8192 16384 24576 32768 49152 65536 98304 131072 262144; do
  echo "channels $c:"; timeout 120 python3 tools/run_once.py $c 2 3 2>&1 | grep "rep 2"
done
echo; echo "### the same with SAME_BATCH_RELAXED (SAME_RELAXED=1: the pipeline's FASTMATH build up to 32 768 channels, the one- / two-wavefront relaxed kernel beyond)"
for c in 4096 16384 32768 49152 65536 98304 131072 262144; do
  echo "channels $c:"; SAME_RELAXED=1 timeout 120 python3 tools/run_once.py $c 2 3 2>&1 | grep "rep 2"
done
echo; echo "### 65 536 / 131 072 channels, relaxed, one wavefront per 64 channels forced (SAME_RELAXED_KERNEL=solo) and two (duo)"
for k in solo duo; do for c in 65536 131072; do echo "$k, channels $c:"; SAME_RELAXED=1 SAME_RELAXED_KERNEL=$k timeout 120 python3 tools/run_once.py $c 2 3 2>&1 | grep "rep 2"; done; done
echo; echo "### 44.1 kHz and 48 kHz, 16 384 channels x 2 s; configs[2] at full length (16 384 ch x 10 s at 48 kHz)"
timeout 120 python3 tools/run_once.py 16384 2 3 44100 2>&1 | grep "rep 2"
timeout 120 python3 tools/run_once.py 16384 2 3 48000 2>&1 | grep "rep 2"
timeout 200 python3 tools/run_once.py 16384 10 2 48000 2>&1 | grep "rep 1"
echo; echo "### configs[1] variants: strict default, one wavefront per 64 channels (SAME_PIPE=0), time-parallel on a time-major and on a channel-major input (8 ... 12 pieces)"
timeout 120 python3 tools/run_once.py 4096 10 3 2>&1 | grep "rep 2"
SAME_PIPE=0 timeout 200 python3 tools/run_once.py 4096 10 3 2>&1 | grep "rep 2"
SAME_TP=1 timeout 120 python3 tools/run_once.py 4096 10 3 2>&1 | grep "rep 2"
for k in 8 9 10 11 12; do echo "pieces $k:"; TP_CHUNKS=$k SAME_TP_SORT=1 timeout 120 python3 tools/tp_cm_once.py 4096 10 4 2>&1 | grep "rep 3"; done
echo "8 pieces, grid order (round 2's form of the launch):"; TP_CHUNKS=8 SAME_TP_SORT=0 timeout 120 python3 tools/tp_cm_once.py 4096 10 4 2>&1 | grep "rep 3"
} > $OUT/configs.txt 2>&1
cd /tmp && export TMPDIR=/tmp
SAME_RELAXED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace262k -- python3 $R/tools/run_once.py 262144 2 3 > $OUT/trace262k.log 2>&1
find $OUT/trace262k -name "*kernel_stats.csv" -newer $OUT/configs.txt -exec head -6 {} \; >> $OUT/configs.txt
cat $OUT/configs.txt
