#!/bin/bash
# usage (GPU box): tools/trace_rounds.sh <workload> <scale> : per-launch durations of the solver-round kernels of the longest solve
w=$1; sc=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/trr
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/trr -- python /root/repo/bench.py --workload $w --scale $sc --steps 1 --warmup 1 --no-cpu-baseline --groups-in-flight 1 "$@" > /dev/null 2>&1
cd /root/repo
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trr/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
solves, cur = [], []
for r in rows:
    n = r["Kernel_Name"]
    if n.startswith("gf_build"):
        cur = []
        solves.append(cur)
    if "gf_count_claim" in n or "gf_check_apply" in n:
        cur.append((("C" if "count" in n else "A"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000))
best = max(solves[len(solves) // 2:], key=lambda s: sum(d for _, d in s))
print("rounds of the longest solve (us):", " ".join("%s%.0f" % x for x in best))
print("total %.1f ms over %d launches" % (sum(d for _, d in best) / 1000, len(best)))
PY
