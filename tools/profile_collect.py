#!/usr/bin/env python3
"""Turn gpurun_out/prof (written by tools/profile_round.sh on the GPU box) into the files under profiles/:
bench line, rocprofv3 kernel/domain stats, per-launch durations of the demod kernel, the two PMC
passes restricted to the demod kernel, and r01_traffic.json (what bench.py reports as roofline.traffic).

    python tools/profile_collect.py [round-prefix, default r01]

FETCH_SIZE correction: 1.0 for the 16-channel workgroup shape, 2.0 for 64 channels per wavefront
(profiles/r01_fetch_calibration.txt); picked from the kernel's template arguments."""
import csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles")
R = sys.argv[1] if len(sys.argv) > 1 else "r01"


def one(pattern):
    # gpurun merges new files into gpurun_out/ without removing those of earlier calls: take the newest
    return max(glob.glob(os.path.join(P, pattern)), key=os.path.getmtime)


def demod_rows(d):
    rows = list(csv.reader(open(one(f"{d}/*/*_counter_collection.csv"))))
    return [rows[0]] + [r for r in rows[1:] if "demod" in r[8]]


shutil.copy(os.path.join(P, "bench.json"), os.path.join(OUT, f"{R}_bench_4096ch_10s.json"))
shutil.copy(one("trace/*/*_kernel_stats.csv"), os.path.join(OUT, f"{R}_rocprofv3_kernel_stats.csv"))
shutil.copy(one("trace/*/*_domain_stats.csv"), os.path.join(OUT, f"{R}_rocprofv3_domain_stats.csv"))
per = {}
for d, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = demod_rows(d)
    with open(os.path.join(OUT, f"{R}_pmc_{name}.csv"), "w", newline="") as f:
        csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(rows)
    vals = [float(r[16]) for r in rows[1:]]
    per[name] = (sum(vals) / len(vals), len(vals), rows[1][8])
kernel = per["FETCH_SIZE"][2].split("(")[0].replace("void same::", "")
m = re.search(r"demod_pipe_kernel<\d+, \d+, \d+, \w+, \w+, (\d+)", kernel)
corr = 1.0 if (m and m.group(1) == "16") else 2.0
bench = json.loads(open(os.path.join(P, "bench.json")).read().strip().splitlines()[-1])
C, T = bench["config"]["channels_per_gpu"], bench["config"]["samples_per_channel"]
traffic = int(round(per["FETCH_SIZE"][0] * 1024 * corr + per["WRITE_SIZE"][0] * 1024))
json.dump({"workload": f"{C} ch x {T} samples", "hbm_bytes_per_launch": traffic,
           "fetch_size_kb": round(per["FETCH_SIZE"][0], 1), "write_size_kb": round(per["WRITE_SIZE"][0], 1),
           "fetch_correction": corr,
           "source": f"profiles/{R}_pmc_FETCH_SIZE.csv, {R}_pmc_WRITE_SIZE.csv ({per['FETCH_SIZE'][1]} launches each of {kernel}); "
                     f"FETCH_SIZE counts this kernel's wavefront loads x{corr:g}: r01_fetch_calibration.txt"},
          open(os.path.join(OUT, f"{R}_traffic.json"), "w"), indent=1)
d = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
           for r in csv.DictReader(open(one("trace/*/*_kernel_trace.csv"))) if "demod_pipe_kernel" in r["Kernel_Name"] or "demod_fast_kernel" in r["Kernel_Name"])
log = open(os.path.join(P, "trace_bench.log")).read()
live = re.search(r'"kernel_ms": ([0-9.]+)', log)
warm = len(d) - bench["steps"]
with open(os.path.join(OUT, f"{R}_rocprofv3_kernel_trace_demod.txt"), "w") as o:
    o.write(f"{kernel} launches of\n`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-scaled --check 0`\n"
            f"(tools/profile_round.sh; {warm} warm-up + {bench['steps']} timed; durations in ms, in launch order):\n")
    o.write(" ".join(f"{x[1]:.3f}" for x in d) + "\n")
    o.write(f"average of the {bench['steps']} timed launches: {sum(x[1] for x in d[warm:]) / bench['steps']:.3f} ms   "
            f"(bench.py's HIP-event figure for the same launches, printed by that run: {live.group(1) if live else '?'} ms)\n")
    o.write(f"average of all {len(d)} (what *_kernel_stats.csv reports): {sum(x[1] for x in d) / len(d):.3f} ms\n")
    o.write(f"An unprofiled `python bench.py` on the same box reported kernel_ms {bench['roofline']['kernel_ms']} ({R}_bench_4096ch_10s.json).\n")
print(open(os.path.join(OUT, f"{R}_rocprofv3_kernel_trace_demod.txt")).read())
print(json.dumps(json.load(open(os.path.join(OUT, f"{R}_traffic.json"))), indent=1))
print("algorithmic", 4 * C * T, "ratio", traffic / (4.0 * C * T))
print("bench value", bench["value"], "ms_per_step", bench["ms_per_step"], "frac", bench["roofline"]["frac"])
