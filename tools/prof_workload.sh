#!/bin/bash
# usage (GPU box): tools/prof_workload.sh <workload> <scale> <tag>
w=$1; sc=$2; tag=$3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/pw_$tag -- python /root/repo/bench.py --workload $w --scale $sc --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pw_$tag.json 2>/dev/null
cd /root/repo
python - <<PY
import csv,glob,json
f=glob.glob('gpurun_out/pw_$tag/*/*kernel_stats.csv')[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<16: print(r['Name'][:44].ljust(44), r['Calls'].rjust(5), r['TotalDurationNs'].rjust(11), r['AverageNs'][:10].rjust(11), r['MaxNs'].rjust(10))
d=json.load(open('gpurun_out/pw_$tag.json'))
print(d['ms_per_step'], d['kernel_ms_per_step'], d['work_per_step'])
PY
