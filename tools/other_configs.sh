#!/bin/bash
# Kernel times (HIP events, tools/run_once.py) at the configurations the bench line does not carry, plus a rocprofv3
# kernel trace of the largest one:   bash tools/other_configs.sh  -> gpurun_out/other/configs.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/other
mkdir -p $OUT
cd $R
{
echo "### channel-count sweep at 22.05 kHz, 2 s per launch (kernel chosen by the dispatcher; dense = one-wavefront kernel built for two wavefronts per SIMD)"
for c in 1024 2048 4096 8192 16384 24576 32768 49152 65536 98304 131072 262144; do
  echo "channels $c:"; python3 tools/run_once.py $c 2 3 2>&1 | grep "rep 2"
done
echo; echo "### the same three largest without the dense build (SAME_FAST_DENSE=0)"
for c in 98304 131072 262144; do echo "channels $c:"; SAME_FAST_DENSE=0 python3 tools/run_once.py $c 2 3 2>&1 | grep "rep 2"; done
echo; echo "### 44.1 kHz and 48 kHz, 16 384 channels x 2 s; configs[2] at full length (16 384 ch x 10 s at 48 kHz)"
python3 tools/run_once.py 16384 2 3 44100 2>&1 | grep "rep 2"
python3 tools/run_once.py 16384 2 3 48000 2>&1 | grep "rep 2"
python3 tools/run_once.py 16384 10 2 48000 2>&1 | grep "rep 1"
echo; echo "### configs[1] variants: strict default, one wavefront per 64 channels (SAME_PIPE=0), time-parallel"
python3 tools/run_once.py 4096 10 3 2>&1 | grep "rep 2"
SAME_PIPE=0 python3 tools/run_once.py 4096 10 3 2>&1 | grep "rep 2"
python3 tools/tp_cm_once.py 4096 10 3 2>&1 | grep "rep 2"
} > $OUT/configs.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace262k -- python3 $R/tools/run_once.py 262144 2 3 > $OUT/trace262k.log 2>&1
find $OUT/trace262k -name "*kernel_stats.csv" -exec head -6 {} \; >> $OUT/configs.txt
tail -12 $OUT/configs.txt
