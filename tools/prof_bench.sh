#!/bin/bash
# usage (on the GPU box): tools/prof_bench.sh <tag>  -> gpurun_out/prof_<tag>/ + compact kernel table
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$tag -- python /root/repo/bench.py --steps 10 --warmup 2 --no-overlap-figure > /root/repo/gpurun_out/prof_$tag.json 2>/dev/null
cd /root/repo
python - <<PY
import csv,glob,json
f=glob.glob('gpurun_out/prof_$tag/*/*kernel_stats.csv')[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<22: print(r['Name'][:40].ljust(40), r['Calls'].rjust(5), r['TotalDurationNs'].rjust(10), r['AverageNs'][:9].rjust(10), r['MinNs'].rjust(7), r['MaxNs'].rjust(8))
d=json.load(open('gpurun_out/prof_$tag.json'))
print(d['ms_per_step'], d['kernel_ms_per_step'], d['parity_vs_oracle'])
PY
