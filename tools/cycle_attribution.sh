#!/bin/bash
# Shader-clock attribution of the pipeline kernel's wavefronts, on the GPU box (through gpurun):
#   bash tools/cycle_attribution.sh   -> gpurun_out/cycle/attribution.txt   (filed as profiles/rNN_cycle_attribution.txt)
# Builds the SAME_PROFILE=1 library (never the shipped one: it stamps the shader clock at section boundaries of
# workgroup 0's wavefronts) and runs tools/run_once.py over the two regimes of the bench line, plain and with
# parts of the pipeline knocked out (SAME_PIPE_PRIO bits, same_profile.h: the results are garbage then, the
# timing says what the step waits for).  The normal build is restored at the end.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cycle
mkdir -p $OUT
A=$OUT/attribution.txt
cd $R
export SAME_PROFILE=1
python3 -m sameold_amd.build > /dev/null 2>&1
F='rep 1\|stage \|DC wave\|polled\|second\|HW_ID'
{
echo "### configs[1]: 4 096 channels x 10 s, 16-channel workgroups, five wavefronts (clk per 20-sample step of workgroup 0)"
python3 tools/run_once.py 4096 10 2 2>&1 | grep "$F"
for p in 16 440 504; do
  echo; echo "### configs[1], knock-out mask $p (8 helper's events, 16 symbol path, 32 stage 2, 64 AGC, 128 DC blocker, 256 helper's filters)"
  SAME_PIPE_PRIO=$p python3 tools/run_once.py 4096 10 2 2>&1 | grep "rep 1\|stage \|DC wave"
done
echo; echo "### full-chip regime: 32 768 channels x 2 s, 64-channel workgroups, two per CU, four wavefronts each"
python3 tools/run_once.py 32768 2 2 2>&1 | grep "$F"
for p in 8 16 32 64 256 504; do
  echo; echo "### 32 768 channels, knock-out mask $p"
  SAME_PIPE_PRIO=$p python3 tools/run_once.py 32768 2 2 2>&1 | grep "rep 1\|stage [1-4]"
done
echo; echo "### time-parallel launch, configs[1] channel-major (fractions of workgroup 0's time)"
python3 tools/tp_cm_once.py 4096 10 3 2>&1 | tail -4
} > $A 2>&1
export SAME_P3_MARKS=1
python3 -m sameold_amd.build > /dev/null 2>&1
{
echo; echo "### symbol path (stage 3) by section, SAME_P3_MARKS=1 build: configs[1]"
python3 tools/run_once.py 4096 10 2 2>&1 | grep "rep 1\|stage 3"
echo; echo "### symbol path by section: 32 768 channels"
python3 tools/run_once.py 32768 2 2 2>&1 | grep "rep 1\|stage 3"
} >> $A 2>&1
unset SAME_PROFILE SAME_P3_MARKS
python3 -m sameold_amd.build > /dev/null 2>&1
tail -5 $A
