#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/pytest6.log
tail -5 gpurun_out/r3/pytest6.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/pv_c09 -- python /root/repo/tools/c09_bench.py 0.9 0,7 > /root/repo/gpurun_out/r3/c09prof.log 2>&1
cd /root/repo
tail -3 gpurun_out/r3/c09prof.log
python - <<'PY'
import csv,glob,os
f=sorted(glob.glob('gpurun_out/pv_c09/*/*kernel_stats.csv'), key=os.path.getmtime)[-1]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<22: print(r['Name'][:46].ljust(46), r['Calls'].rjust(6), ("%.2f"%(float(r['TotalDurationNs'])/1e6)).rjust(9), "ms", r['AverageNs'][:9].rjust(10))
PY
