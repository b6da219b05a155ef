cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
rm -rf $R/gpurun_out/tptrace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tptrace -- python3 $R/tools/tp_cm_once.py 4096 10 5 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
f=max(glob.glob(os.environ["GRAFT_REPO_ROOT"]+'/gpurun_out/tptrace/*/*_kernel_trace.csv'),key=os.path.getmtime)
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if 'demod_pipe_kernel' in r['Kernel_Name']]
prev,i0=idx[-2],idx[-1]
t0=int(rows[prev]['End_Timestamp'])
for r in rows[prev+1:i0+4]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  {r['Kernel_Name'][:50]}")
PY
