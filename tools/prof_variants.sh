#!/bin/bash
# usage (GPU box): tools/prof_variants.sh <tag> "<ENV=1 ...>" [bench args]  -> top kernels of an S4 bench run under rocprofv3
tag=$1; envs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/pv_$tag -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure "$@" > /root/repo/gpurun_out/pv_$tag.json 2>/root/repo/gpurun_out/pv_$tag.err
cd /root/repo
TAG=$tag python - <<'PY'
import csv,glob,json,os
tag=os.environ["TAG"]
f=sorted(glob.glob('gpurun_out/pv_%s/*/*kernel_stats.csv'%tag), key=os.path.getmtime)[-1]
print("==", tag)
for i,r in enumerate(csv.DictReader(open(f))):
    if i<14: print(r['Name'][:40].ljust(40), r['Calls'].rjust(5), ("%.2f"%(float(r['TotalDurationNs'])/4e6)).rjust(9), "ms/step", r['AverageNs'][:9].rjust(10))
try:
    d=json.loads(open('gpurun_out/pv_%s.json'%tag).read().strip().splitlines()[-1])
    print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if isinstance(v,float)}, d.get('parity_vs_golden_digests'))
    w=d['work_per_step']; print({k:w[k] for k in ('rows_recounted','bitmap_words_read','flat_rows_streamed','flat_rows_recounted','flat_owner_words')})
except Exception as e:
    print("json", e, open('gpurun_out/pv_%s.err'%tag).read()[-500:])
PY
