#!/bin/bash
# usage (GPU box): tools/pmc_kernel.sh <workload> <scale> <kernel-substring> "<counters...>" [more bench args]
w=$1; sc=$2; kern=$3; ctrs=$4; shift 4
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/pmck
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmck -- python /root/repo/bench.py --workload $w --scale $sc --steps 1 --warmup 1 --no-cpu-baseline --groups-in-flight 1 "$@" > /dev/null 2>&1
cd /root/repo
KERN="$kern" python - <<'PY'
import csv, glob, collections, os
f = glob.glob("gpurun_out/pmck/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if os.environ["KERN"] not in r["Kernel_Name"]: continue
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(r["Kernel_Name"][:40], r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s total %.4g  per-dispatch %.4g  (%d dispatches)" % (c, v, v / n[(k, c)], n[(k, c)]))
PY
