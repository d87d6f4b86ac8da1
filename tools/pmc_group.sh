#!/bin/bash
# usage (GPU box): tools/pmc_group.sh <group> <kernel-substring> "<counters...>"  : PMC counters of one kernel on one S4 group
g=$1; kern=$2; ctrs=$3
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/pmcg
S4_GROUPS=$g rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmcg -- python /root/repo/tools/s4_groups.py S4 1.0 1 > /dev/null 2>&1
cd /root/repo
KERN="$kern" python - <<'PY'
import csv, glob, collections, os
f = glob.glob("gpurun_out/pmcg/*/*counter_collection.csv")[0]
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
