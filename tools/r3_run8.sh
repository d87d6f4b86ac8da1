#!/bin/bash
mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/pv_s5 -- python /root/repo/bench.py --workload S5 --scale 0.25 --steps 1 --warmup 0 --no-solver-check > /root/repo/gpurun_out/r3/s5_prof.json 2>/root/repo/gpurun_out/r3/s5_prof.err
cd /root/repo
python - <<'PY'
import csv,glob,os,json
f=sorted(glob.glob('gpurun_out/pv_s5/*/*kernel_stats.csv'), key=os.path.getmtime)[-1]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<25: print(r['Name'][:50].ljust(50), r['Calls'].rjust(7), ("%.1f"%(float(r['TotalDurationNs'])/1e6)).rjust(9), "ms", r['AverageNs'][:9].rjust(10))
d=json.loads(open('gpurun_out/r3/s5_prof.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['wall_s_per_step'])
PY
