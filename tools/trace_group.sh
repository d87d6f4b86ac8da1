#!/bin/bash
# usage (GPU box): tools/trace_group.sh <group> [scale] : per-launch kernel durations of one S4 group's fused filter call
g=${1:-0}; sc=${2:-1.0}
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/trg
S4_GROUPS=$g rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/trg -- python /root/repo/tools/s4_groups.py S4 $sc 2 > /root/repo/gpurun_out/trg.txt 2>/dev/null
cd /root/repo
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trg/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last call = from the last seed_init on
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("seed_init")]
rows = rows[idx[-1]:]
t0 = int(rows[0]["Start_Timestamp"]); prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1000 if prev else 0
    print("%9.1f %-44s dur %9.1f gap %7.1f  grid %s wg %s" % ((s - t0) / 1000, r["Kernel_Name"][:44], (e - s) / 1000, gap, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
    prev = e
print(open("gpurun_out/trg.txt").read()[:800])
PY
