#!/bin/bash
# usage (GPU box): tools/pmc_ndf.sh "<counters>"  : PMC counters of the ndf_lazy_kernel dispatches of the first chunks of S5 x 1.0
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/pmcn
CATCHHIP_FRONT_END_WORKERS=1 CATCHHIP_PREFETCH_DEPTH=0 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmcn -- python /root/repo/tools/s5_profile.py 0.25 once > /dev/null 2>&1
cd /root/repo
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmcn/*/*counter_collection.csv")[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "ndf_lazy_kernel" in r["Kernel_Name"] or "ndf_wake" in r["Kernel_Name"]:
        nm = r["Kernel_Name"]
        per[int(r["Dispatch_Id"])]["kind"] = "wake" if "wake" in nm else ("drain" if "true, true" in nm else "pass")
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        per[int(r["Dispatch_Id"])]["grid"] = float(r.get("Grid_Size", 0) or 0)
for k, (d, c) in enumerate(sorted(per.items())):
    if k < 90 and (k < 12 or k % 9 < 3):
        print(k, c.pop("kind"), "  ".join("%s=%.4g" % kv for kv in sorted(c.items())))
PY
