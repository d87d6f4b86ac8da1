#!/bin/bash
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/prof35; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python /root/repo/bench.py --workload S4 --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure > $out/bench.json 2>/dev/null
f=$(ls -t $out/trace/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv; rm -rf $out/trace
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/prof35/kernel_stats.csv')))
for r in rows[:16]:
    print(r['Name'][:45].ljust(45), r['Calls'].rjust(6), round(int(r['TotalDurationNs'])/1e6/7,2))
PY
