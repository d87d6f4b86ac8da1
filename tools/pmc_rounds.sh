#!/bin/bash
# usage (GPU box): tools/pmc_rounds.sh <group> "<counters...>" [n]  : PMC counters of the first n (6) gr_count / gr_claim
# dispatches of one S4 group's solve, dispatch by dispatch (one solve: reps = 1)
g=$1; ctrs=$2; nd=${3:-6}
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/pmcr
S4_GROUPS=$g rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmcr -- python /root/repo/tools/s4_groups.py S4 1.0 1 > /dev/null 2>&1
cd /root/repo
ND=$nd python - <<'PY'
import csv, glob, collections, os
f = glob.glob("gpurun_out/pmcr/*/*counter_collection.csv")[0]
nd = int(os.environ["ND"])
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    kn = r["Kernel_Name"]
    if "gr_count" in kn or "gr_claim" in kn:
        per[(int(r["Dispatch_Id"]), kn[:24])][r["Counter_Name"]] = float(r["Counter_Value"])
seen = collections.Counter()
for (d, kn), c in sorted(per.items()):
    seen[kn] += 1
    if seen[kn] <= nd:
        print(kn, seen[kn] - 1, "  ".join("%s=%.4g" % kv for kv in sorted(c.items())))
PY
