#!/usr/bin/env python3
"""Turn gpurun_out/prof (written by tools/profile_round.sh on the GPU box) into the files under profiles/:
bench line, rocprofv3 kernel/domain stats, per-launch durations of every demod kernel variant the bench
runs, the PMC passes restricted to them, and rNN_traffic.json (what bench.py reports as roofline.traffic).

    python tools/profile_collect.py [round-prefix, default r03]

FETCH_SIZE correction: 1.0 for the 16-channel workgroup shape (64-byte rows per wavefront load), 2.0 for
64 channels per wavefront (profiles/r01_fetch_calibration.txt); picked from the kernel's template arguments."""
import collections, csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles")
R = sys.argv[1] if len(sys.argv) > 1 else "r05"


def one(pattern):
    # gpurun merges new files into gpurun_out/ without removing those of earlier calls: take the newest
    return max(glob.glob(os.path.join(P, pattern)), key=os.path.getmtime)


def short(name):
    return name.split("(")[0].replace("void same::", "")


def classify(name, wgs, state, n1):
    """Which block of the bench line a demodulation launch belongs to, from the kernel's template arguments
    <NT, NFF, NFB, MED3, SHARE, LANES, SPLIT, SampleT, FASTMATH, input form> and its workgroup count (bench.py runs the blocks in a fixed
    order; `state` counts what has been seen, n1 = launches per configs[1] block)."""
    if "demod_relaxed_kernel" in name or "demod_duo_kernel" in name:
        return "scaled_big_relaxed"
    if "demod_sym_kernel" in name:
        # <NT, NFF, NFB, SampleT, input form> (round 6: NT first): the symbol-paced pipeline takes every relaxed launch; at 22.05 kHz a
        # workgroup is twelve wavefronts = two groups of 64 state columns, at 44.1 / 48 kHz six wavefronts = one group
        m = re.search(r"demod_sym_kernel<(\d+), \d+, \d+, \w+, (\d)>", name)
        if m and int(m.group(1)) != 42:
            return "configs2_48k_relaxed"
        cols = wgs * 128
        if m and m.group(2) == "1":
            return "time_parallel"                             # per-lane input streams: the channel-major time-parallel launch (8 pieces: 32 768 columns)
        if cols == C:
            return "relaxed"
        if cols == 131072:
            return "scaled_big_relaxed"
        state["sym512"] = state.get("sym512", 0) + 1          # 32 768 columns, time-major rows: configs[1] cut uniformly first, the shard later
        return "time_parallel_time_major" if state["sym512"] <= n1 else "scaled_relaxed"
    m = re.search(r"demod_pipe_kernel<(\d+), \d+, \d+, \w+, (\w+), (\d+), \w+, \w+, (\w+)(?:, \d+)?>", name)
    if not m:
        return None
    nt, share, lanes, fm = int(m.group(1)), m.group(2) in ("true", "1"), int(m.group(3)), m.group(4) in ("true", "1")
    if nt == 92:
        return "configs2_48k_relaxed" if fm else "configs2_48k"
    if lanes == 16:
        return "strict"
    cols = wgs * 64
    if fm and cols == C:
        return "relaxed"
    if cols > 32768:
        return "time_parallel" if fm else "time_parallel_strict_chunks"
    if fm:          # 32 768 columns: configs[1] cut uniformly on the time-major buffer first, the 32 768-channel shard later
        state["fm512"] = state.get("fm512", 0) + 1
        return "time_parallel_time_major" if state["fm512"] <= n1 else "scaled_relaxed"
    return "scaled"


bench = json.loads(open(os.path.join(P, "bench.json")).read().strip().splitlines()[-1])
shutil.copy(os.path.join(P, "bench.json"), os.path.join(OUT, f"{R}_bench_4096ch_10s.json"))
shutil.copy(one("trace/*/*_kernel_stats.csv"), os.path.join(OUT, f"{R}_rocprofv3_kernel_stats.csv"))
shutil.copy(one("trace/*/*_domain_stats.csv"), os.path.join(OUT, f"{R}_rocprofv3_domain_stats.csv"))
C, T = bench["config"]["channels_per_gpu"], bench["config"]["samples_per_channel"]
CONFIGS1 = ("strict", "time_parallel", "time_parallel_time_major", "time_parallel_strict_chunks", "relaxed")

# ---- per-launch durations, in launch order, per block
DEMOD = ("demod_pipe_kernel", "demod_fast_kernel", "demod_relaxed_kernel", "demod_duo_kernel", "demod_sym_kernel")
all_rows = list(csv.DictReader(open(one("trace/*/*_kernel_trace.csv"))))
all_rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in all_rows if any(k in r["Kernel_Name"] for k in DEMOD)]
blocks = collections.OrderedDict()
state = {}
n1_trace = bench["steps"] + bench["warmup"] + 1          # (+ the untimed pass after the timed ones)
try:
    # the traced run's own line says how many launches its blocks made (pre-heat passes run by the clock, not by count)
    traced = json.loads([l for l in open(os.path.join(P, "trace_bench.log")).read().splitlines() if l.startswith('{"metric"')][-1])
    n1_trace = int(traced["modes"]["time_parallel_time_major"]["launches"])
except Exception:
    pass
for r in rows:
    wgs = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    v = classify(r["Kernel_Name"], wgs, state, n1_trace) or short(r["Kernel_Name"])
    blocks.setdefault(v, []).append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, short(r["Kernel_Name"]), wgs))
# the small kernels a time-parallel launch brackets with its HIP events (scout, planner, sort, state columns)
side = collections.defaultdict(list)
for r in all_rows:
    n = short(r["Kernel_Name"]).replace("same::", "")
    if n.startswith(("tp_", "copy_state_columns", "init_state", "chunk_final", "fill_u64")):
        n = f"{n} x{int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)}"
        side[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
log = open(os.path.join(P, "trace_bench.log")).read()
with open(os.path.join(OUT, f"{R}_rocprofv3_kernel_trace_demod.txt"), "w") as o:
    o.write("demodulation kernel launches of\n`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --check 0 --no-carried-state-check`\n"
            "(tools/profile_round.sh), durations in ms in launch order, per block of the bench line (workgroups = grid / workgroup size):\n")
    for v, lst in blocks.items():
        n_timed = bench["steps"] if v in CONFIGS1 else max(bench["steps"], 10)
        d = [x[0] for x in lst]
        extra = 1 if v in CONFIGS1 else 0       # the last launch of a configs[1] block is the untimed pass after the timed ones
        timed = d[len(d) - extra - n_timed:len(d) - extra] if len(d) > extra + n_timed - 1 else d
        o.write(f"\n[{v}] {lst[0][1]}  {lst[0][2]} workgroups  ({len(d)} launches)\n  " + " ".join(f"{x:.3f}" for x in d) + "\n")
        o.write(f"  average of the {len(timed)} timed ones: {sum(timed) / max(len(timed), 1):.3f} ms; of all: {sum(d) / len(d):.3f} ms\n")
    o.write("\nkernels around a time-parallel launch (name x workgroups; microseconds: shortest / average over their launches).  Scout,\n"
            "planner and sort run on the plan stream BESIDE the previous launch's tail when the caller keeps two calls in flight (their\n"
            "durations then stretch -- they wait for CUs the tail still holds -- and are hidden); the shortest figure is what they take alone:\n")
    for n, d in sorted(side.items()):
        o.write(f"  {n:40s} {min(d):8.1f} / {sum(d) / len(d):8.1f} us  x {len(d)}\n")
    m = re.findall(r'"kernel_ms": ([0-9.]+)', log)
    o.write(f"\nbench.py's own HIP-event figures printed by that profiled run (kernel_ms, in the order of the JSON line): {' '.join(m)}\n")
    o.write(f"An unprofiled `python bench.py` on the same box: {R}_bench_4096ch_10s.json.\n")

# ---- PMC: FETCH_SIZE / WRITE_SIZE per block
def counters(d):
    rows = list(csv.DictReader(open(one(f"{d}/*/*_counter_collection.csv"))))
    return [r for r in rows if "demod" in r["Kernel_Name"]]

traffic = []
per = collections.defaultdict(dict)
for d, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = counters(d)
    with open(os.path.join(OUT, f"{R}_pmc_{name}.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader(); w.writerows(rows)
    seen = collections.defaultdict(list)
    state = {}
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
        v = classify(r["Kernel_Name"], wgs, state, 4) or short(r["Kernel_Name"])     # --steps 2 --warmup 1: four launches per configs[1] block
        seen[v].append(float(r["Counter_Value"]))
    for v, vals in seen.items():
        per[v][name] = (sum(vals) / len(vals), len(vals))
sizes = {"strict": (C, T), "time_parallel": (C, T), "time_parallel_time_major": (C, T), "time_parallel_strict_chunks": (C, T), "relaxed": (C, T),
         "scaled": (32768, 44100), "scaled_relaxed": (32768, 44100), "configs2_48k": (16384, 96000), "configs2_48k_relaxed": (16384, 96000),
         "scaled_big_relaxed": (131072, 44100)}
for v, d in per.items():
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d or v not in sizes:
        continue
    # 1.0 for the 64-byte rows of the 16-channel workgroups, 2.0 for 256-byte rows and for the per-lane 16-byte streams
    # of the channel-major time-parallel launch (128-byte requests counted as 64: profiles/r01_fetch_calibration.txt,
    # profiles/r02_fetch_calibration.txt)
    corr = 1.0 if v == "strict" else 2.0
    cc, tt = sizes[v]
    hbm = int(round(d["FETCH_SIZE"][0] * 1024 * corr + d["WRITE_SIZE"][0] * 1024))
    traffic.append({"mode": v if v in CONFIGS1 else ("relaxed" if v.endswith("_relaxed") else "strict"), "block": v, "workload": f"{cc} ch x {tt} samples",
                    "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": 4 * cc * tt, "ratio": round(hbm / (4.0 * cc * tt), 4),
                    "fetch_size_kb": round(d["FETCH_SIZE"][0], 1), "write_size_kb": round(d["WRITE_SIZE"][0], 1), "fetch_correction": corr,
                    "launches": d["FETCH_SIZE"][1],
                    "source": f"profiles/{R}_pmc_FETCH_SIZE.csv, {R}_pmc_WRITE_SIZE.csv; FETCH_SIZE counts this kernel shape's wavefront loads x{corr:g}: r01_fetch_calibration.txt"})
json.dump(traffic, open(os.path.join(OUT, f"{R}_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(P, "pmc_summary.txt"), os.path.join(OUT, f"{R}_pmc_summary.txt"))

# ---- instruction mix per block of the bench line (the SQ pass: --steps 2 --warmup 1, three launches per block)
try:
    rows = counters("sq")
    mix = collections.OrderedDict()
    seen_disp = collections.defaultdict(set)
    state = {}
    by_disp = {}
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        if r["Dispatch_Id"] not in by_disp:
            wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            by_disp[r["Dispatch_Id"]] = classify(r["Kernel_Name"], wgs, state, 4) or short(r["Kernel_Name"])
        v = by_disp[r["Dispatch_Id"]]
        seen_disp[v].add(r["Dispatch_Id"])
        mix.setdefault(v, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    lanes = {"strict": 16}
    with open(os.path.join(OUT, f"{R}_pmc_instruction_mix.txt"), "w") as o:
        o.write("Wavefront instructions per input sample of one workgroup (its 16 or 64 channels advance one sample), per block of the\n"
                "bench line: `rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY\n"
                "-- python3 bench.py --no-cpu-baseline --check 0 --steps 2 --warmup 1` (tools/profile_round.sh), summed over the launches of a\n"
                "block.  Time-parallel blocks: per sample of the INPUT (their columns also process warm-ups and run-ons).\n\n")
        for v, c in mix.items():
            if v not in sizes:
                continue
            cc, tt = sizes[v]
            n = len(seen_disp[v])
            wg = lanes.get(v, 64)
            per = cc / wg * tt * n                      # workgroup-samples over the block's launches
            o.write(f"[{v}] {cc} ch x {tt} samples, {n} launches, {wg}-channel workgroups\n")
            o.write(f"  VALU {c['SQ_INSTS_VALU'] / per:7.1f}   SALU {c['SQ_INSTS_SALU'] / per:6.1f}   LDS {c['SQ_INSTS_LDS'] / per:6.1f}   per workgroup-sample"
                    f"  (= {c['SQ_INSTS_VALU'] / per / wg * 64 / 64:.2f} VALU per channel-sample x {wg})\n")
            wc = c["SQ_WAVE_CYCLES"]
            o.write(f"  wave cycles {wc / max(c['SQ_WAVES'], 1):.3e} per wavefront; VALU issuing {c['SQ_ACTIVE_INST_VALU'] / wc:.1%} of wave cycles, waiting (any) {c['SQ_WAIT_ANY'] / wc:.1%}\n\n")
    # what bench.py quotes beside roofline.frac: vector instructions per 64-column sample and how busy a SIMD's vector unit is
    # (instructions x 4 clk, over the launch's cycles on 1 024 SIMDs: SQ_WAVE_CYCLES counts per wavefront, a SIMD holds
    # waves_per_simd of them)
    issue = []
    for v, c in mix.items():
        if v not in sizes:
            continue
        cc, tt = sizes[v]
        n = len(seen_disp[v])
        wg = lanes.get(v, 64)
        per = cc / wg * tt * n
        wc, waves = c["SQ_WAVE_CYCLES"], max(c["SQ_WAVES"], 1)
        issue.append({"mode": v if v in CONFIGS1 else ("relaxed" if v.endswith("_relaxed") else "strict"), "block": v, "workload": f"{cc} ch x {tt} samples",
                      "valu_per_workgroup_sample": round(c["SQ_INSTS_VALU"] / per, 2), "salu_per_workgroup_sample": round(c["SQ_INSTS_SALU"] / per, 2),
                      "lds_per_workgroup_sample": round(c["SQ_INSTS_LDS"] / per, 2),
                      "valu_issue_fraction_of_wave_cycles": round(c["SQ_ACTIVE_INST_VALU"] / wc, 4),
                      "source": f"profiles/{R}_pmc_instruction_mix.txt"})
    json.dump(issue, open(os.path.join(OUT, f"{R}_issue.json"), "w"), indent=1)
    print(open(os.path.join(OUT, f"{R}_pmc_instruction_mix.txt")).read())
except Exception as e:      # an old profile directory without the SQ pass
    print("no instruction mix:", e)
print(open(os.path.join(OUT, f"{R}_rocprofv3_kernel_trace_demod.txt")).read())
print(json.dumps(traffic, indent=1))
print("bench value", bench["value"], "ms_per_step", bench["ms_per_step"], "frac", bench["roofline"]["frac"], "mode", bench["config"].get("mode"))
