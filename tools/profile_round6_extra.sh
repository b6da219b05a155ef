#!/bin/bash
# Round 6's measurements beside tools/profile_round.sh (run from the repo root through gpurun):
#   the SIMD issue microbenchmark, the `scaled_big.relaxed` launch by itself under both clocks, the light
#   timelines at 22.05 / 44.1 / 48 kHz (needs the `tl` build variant: SAME_BUILD_VARIANT=tl SAME_SYM_TL=1 python -m sameold_amd.build),
#   and the relaxed curve beside strict = oracle at 48 kHz.  Every command under its own timeout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6x
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench_simd > $O/ubench_simd.txt 2>&1
timeout 300 python3 $R/tools/big_once.py > $O/big_plain.txt 2>&1
timeout 300 python3 $R/tools/big_once.py strict_first > $O/big_strict_first.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/big_trace -- python3 $R/tools/big_once.py > $O/big_traced.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/big_trace_sf -- python3 $R/tools/big_once.py strict_first > $O/big_traced_sf.txt 2>&1
python3 - > $O/big_trace_summary.txt 2>&1 <<PY
import csv, glob
for d in ("big_trace", "big_trace_sf"):
    for f in glob.glob("$O/%s/*/*_kernel_trace.csv" % d):
        rows = [r for r in csv.DictReader(open(f)) if "demod_sym_kernel" in r["Kernel_Name"] or "demod_fast_kernel" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        print(d, "rocprofv3 kernel-trace durations (ms), in launch order:")
        for name in ("demod_fast_kernel", "demod_sym_kernel"):
            v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if name in r["Kernel_Name"]]
            if v: print("  ", name, " ".join("%.3f" % x for x in v))
PY
rm -rf $O/big_trace $O/big_trace_sf
SAME_LIB_VARIANT=tl timeout 300 python3 $R/tools/sym_probe.py timeline48 > $O/tl48.txt 2>&1
SAME_LIB_VARIANT=tl timeout 300 python3 $R/tools/sym_probe.py timeline > $O/tl22.txt 2>&1
timeout 900 python3 $R/tests/helpers/ber_vs_oracle.py --rate 48000 --kernel pipe --trials 32768 --batches 8 --relaxed --out $O/r06_ber_vs_oracle_48k_relaxed.json > $O/ber48.log 2>&1
tail -2 $O/ber48.log
