#!/bin/bash
# Round 3: shader-clock attribution of the relaxed kernels and of the pipeline's FASTMATH build.  Needs the SAME_PROFILE build:
#   SAME_PROFILE=1 python -m sameold_amd.build      (here; the library travels with the snapshot)
#   gpurun -- 'bash tools/cycle_attribution_r3.sh'  -> gpurun_out/cycle3/attribution.txt  (-> profiles/r03_cycle_attribution.txt)
#   python -m sameold_amd.build                     (back to the shipped build)
# Every run under its own timeout (a knocked-out pipeline stage can hang a hand-over).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cycle3
mkdir -p $OUT
A=$OUT/attribution.txt
cd $R
export SAME_PROFILE=1
{
echo "SAME_PROFILE build (slower than the shipped one by the stamps it takes); clk = shader clock of workgroup 0"
echo
echo "### full-chip regime, 32 768 channels x 2 s, pipeline FASTMATH (SAME_RELAXED=1): clk per 20-sample step"
SAME_RELAXED=1 timeout 120 python3 tools/run_once.py 32768 2 3 2>&1 | grep -v amdgpu | tail -12
echo; echo "### the same, strict pipeline"
timeout 120 python3 tools/run_once.py 32768 2 3 2>&1 | grep -v amdgpu | tail -12
echo; echo "### time-parallel launch of configs[1] (4 096 ch x 220 500, channel-major, 10 pieces), pipeline FASTMATH: workgroup 0"
timeout 120 python3 tools/tp_cm_once.py 4096 10 3 2>&1 | grep -v amdgpu | tail -7
echo; echo "### the same with strict chunks (SAME_RELAXED=0)"
SAME_RELAXED=0 timeout 120 python3 tools/tp_cm_once.py 4096 10 3 2>&1 | grep -v amdgpu | tail -7
for p in 8 16 32 256; do
  echo; echo "### FASTMATH, knock-out mask $p (8 helper's events, 16 symbol path, 32 stage 2, 256 helper's filters)"
  SAME_PIPE_PRIO=$p timeout 120 python3 tools/tp_cm_once.py 4096 10 3 2>&1 | grep -v amdgpu | tail -6
done
for k in solo duo; do
echo; echo "### relaxed kernel ($k), 4 096 channels x 2 s (a wavefront alone on its SIMD)"
SAME_RELAXED_KERNEL=$k timeout 120 python3 tools/relaxed_probe.py prof 4096 2.0 2>&1 | grep -v amdgpu | tail -9
echo; echo "### relaxed kernel ($k), 131 072 channels x 2 s"
SAME_RELAXED_KERNEL=$k timeout 120 python3 tools/relaxed_probe.py prof 131072 2.0 2>&1 | grep -v amdgpu | tail -9
done
} > $A 2>&1
cat $A
