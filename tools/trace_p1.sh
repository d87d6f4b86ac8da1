#!/bin/bash
# usage: tools/trace_p1.sh <tag> : per-round durations of the solver kernels for the first solve of the last step
tag=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/tr_$tag -- python /root/repo/bench.py --steps 3 --warmup 1 > /dev/null 2>&1
cd /root/repo
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/tr_$tag/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=[(r['Kernel_Name'][:22],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000) for r in rows]
idx=[i for i,x in enumerate(n) if x[0].startswith('frow_fill')]
i0=idx[-2] if len(idx)>1 else idx[0]
print("$tag", " ".join(f"{x[0][3:8]}:{x[1]:.0f}" for x in n[i0+1:i0+21]))
PY
