#!/bin/bash
# Round 3: shader-clock attribution of the time-parallel launch (pipeline, strict and FASTMATH, with knock-outs) and of
# the one-wavefront relaxed kernel.  bash tools/cycle_attribution_r3.sh  (through gpurun) -> gpurun_out/cycle3/attribution.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cycle3
mkdir -p $OUT
A=$OUT/attribution.txt
cd $R
export SAME_PROFILE=1
python3 -m sameold_amd.build > /dev/null 2>&1
{
echo "### time-parallel launch of configs[1] (4 096 ch x 220 500, channel-major, 8 chunks), pipeline FASTMATH: workgroup 0"
python3 tools/tp_cm_once.py 4096 10 3 2>&1 | tail -5
echo; echo "### the same with strict chunks (SAME_RELAXED=0)"
SAME_RELAXED=0 python3 tools/tp_cm_once.py 4096 10 3 2>&1 | tail -5
for p in 8 16 32 64 128 256 504; do
  echo; echo "### FASTMATH, knock-out mask $p (8 helper's events, 16 symbol path, 32 stage 2, 64 AGC, 128 DC blocker, 256 helper's filters)"
  SAME_PIPE_PRIO=$p python3 tools/tp_cm_once.py 4096 10 3 2>&1 | tail -5
done
echo; echo "### one-wavefront relaxed kernel, 65 536 channels x 2 s (one wavefront per SIMD, 512-register build)"
python3 tools/relaxed_probe.py prof 65536 2.0 2>&1 | tail -9
echo; echo "### one-wavefront relaxed kernel, 131 072 channels x 2 s (two per SIMD, 256-register build)"
python3 tools/relaxed_probe.py prof 131072 2.0 2>&1 | tail -9
} > $A 2>&1
unset SAME_PROFILE
python3 -m sameold_amd.build > /dev/null 2>&1
cat $A
