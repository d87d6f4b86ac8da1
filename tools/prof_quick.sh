#!/bin/bash
# usage (on the GPU box): tools/prof_quick.sh <tag> [bench args]  -> gpurun_out/prof_<tag>/ + compact kernel table
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$tag -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure "$@" > /root/repo/gpurun_out/prof_$tag.json 2>/root/repo/gpurun_out/prof_$tag.err
cd /root/repo
python - <<PY
import csv,glob,json
f=glob.glob('gpurun_out/prof_$tag/*/*kernel_stats.csv')[0]
print('kernel calls total_ms avg_us min_us max_us')
for i,r in enumerate(csv.DictReader(open(f))):
    if i<40: print(r['Name'][:52].ljust(52), r['Calls'].rjust(6), ('%.2f'%(float(r['TotalDurationNs'])/1e6)).rjust(9), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), ('%.1f'%(float(r['MinNs'])/1e3)).rjust(8), ('%.1f'%(float(r['MaxNs'])/1e3)).rjust(9))
d=json.load(open('gpurun_out/prof_$tag.json'))
print(d['ms_per_step'], d['kernel_ms_per_step'], d.get('parity_vs_golden_digests'))
PY
