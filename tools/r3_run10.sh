#!/bin/bash
mkdir -p /root/repo/gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/pv_s5b -- python /root/repo/tools/s5_profile.py 0.25 > /root/repo/gpurun_out/r3/s5prof3.log 2>&1
cd /root/repo
head -36 gpurun_out/r3/s5prof3.log | cut -c1-150
python - <<'PY'
import csv,glob,os,json
f=sorted(glob.glob('gpurun_out/pv_s5b/*/*kernel_stats.csv'), key=os.path.getmtime)[-1]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<16: print(r['Name'][:50].ljust(50), r['Calls'].rjust(7), ("%.1f"%(float(r['TotalDurationNs'])/2e6)).rjust(9), "ms/step", r['AverageNs'][:9].rjust(10))
PY
