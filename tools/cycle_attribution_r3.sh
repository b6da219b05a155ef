#!/bin/bash
# Round 3: shader-clock attribution of the relaxed kernels (solo: one wavefront per 64 columns; duo: two) and of the
# time-parallel launch of the pipeline.  bash tools/cycle_attribution_r3.sh (through gpurun) -> gpurun_out/cycle3/attribution.txt
# Every run under its own timeout (a knocked-out pipeline stage can hang a hand-over).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cycle3
mkdir -p $OUT
A=$OUT/attribution.txt
cd $R
export SAME_PROFILE=1
timeout 600 python3 -m sameold_amd.build > /dev/null 2>&1
{
for k in solo duo; do
echo "### relaxed kernel ($k), 4 096 channels x 2 s (a wavefront alone on its SIMD)"
SAME_RELAXED_KERNEL=$k timeout 120 python3 tools/relaxed_probe.py prof 4096 2.0 2>&1 | tail -8
echo; echo "### relaxed kernel ($k), 131 072 channels x 2 s"
SAME_RELAXED_KERNEL=$k timeout 120 python3 tools/relaxed_probe.py prof 131072 2.0 2>&1 | tail -8
echo
done
echo "### time-parallel launch of configs[1] (4 096 ch x 220 500, channel-major, 8 chunks), pipeline FASTMATH: workgroup 0"
timeout 120 python3 tools/tp_cm_once.py 4096 10 3 2>&1 | tail -5
echo; echo "### the same with strict chunks (SAME_RELAXED=0)"
SAME_RELAXED=0 timeout 120 python3 tools/tp_cm_once.py 4096 10 3 2>&1 | tail -5
for p in 8 16 32 256; do
  echo; echo "### FASTMATH, knock-out mask $p (8 helper's events, 16 symbol path, 32 stage 2, 256 helper's filters)"
  SAME_PIPE_PRIO=$p timeout 120 python3 tools/tp_cm_once.py 4096 10 3 2>&1 | tail -5
done
} > $A 2>&1
unset SAME_PROFILE
timeout 600 python3 -m sameold_amd.build > /dev/null 2>&1
cat $A
