#!/bin/bash
# usage (GPU box): tools/timeline.sh <tag>  -> kernel timeline of the last solve of the bench
tag=${1:-tl}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/tl_$tag -- python /root/repo/bench.py --steps 3 --warmup 1 --no-overlap-figure $BENCH_ARGS > /root/repo/gpurun_out/tl_$tag.json 2>/dev/null
cd /root/repo
python - <<PY
import csv,glob,json
f=glob.glob('gpurun_out/tl_$tag/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=[(r['Kernel_Name'][:34],int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows]
idx=[i for i,x in enumerate(n) if x[0].startswith('seed_init')]
i0=idx[-1]
prev=None; t0=n[i0][1]
for name,s,e in n[i0:]:
    gap=(s-prev)/1000 if prev else 0
    print(f"{(s-t0)/1000:8.1f} {name:36s} dur {(e-s)/1000:7.1f} gap {gap:7.1f}")
    prev=e
d=json.load(open('gpurun_out/tl_$tag.json'))
print(d['ms_per_step'], d['kernel_ms_per_step'], d['parity_vs_oracle'])
PY
