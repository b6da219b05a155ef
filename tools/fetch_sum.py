import csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"))
rows = [r for r in csv.DictReader(open(f)) if "demod_sym_kernel" in r["Kernel_Name"]]
by = {}
for r in rows: by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in by.items(): print(k, "launches", len(v), "mean KB", sum(v)/len(v), "-> x2 bytes / algorithmic =", sum(v)/len(v)*1024*2/3612672000 if k=="FETCH_SIZE" else sum(v)/len(v)*1024/3612672000)
