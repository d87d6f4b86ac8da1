export CATCHHIP_TEST_HOOKS=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $R/gpurun_out/s5hip -- python $R/tools/s5_profile.py 1.0 > $R/gpurun_out/s5hip.out 2>&1
cd $R
grep "cluster \|stages\|MARK" gpurun_out/s5hip.out
python - <<'PY'
import csv, glob, re, collections
mark = re.search(r"MARK monotonic_ns (\d+) boottime_ns (\d+)", open("gpurun_out/s5hip.out").read())
f = glob.glob("gpurun_out/s5hip/*/*hip_api_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
tmax = max(int(r["End_Timestamp"]) for r in rows)
# which clock?
cands = [int(mark.group(1)), int(mark.group(2))]
m = min(cands, key=lambda c: abs(tmax - c) if c < tmax else 1e30)
print("mark", m, "trace end", tmax, "window s", (tmax - m) / 1e9)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= m]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg[(r["Function"], r["Thread_Id"])]; a[0] += d; a[1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print("%-28s thread %s  %.3f s in %d calls" % (k[0], k[1], v[0] / 1e9, v[1]))
big = sorted(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)[:25]
for r in big:
    print(r["Function"], r.get("Thread_Id"), "%.1f ms at %.3f s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - m) / 1e9))
# kernel busy time per stream (queue) in the window
kf = glob.glob("gpurun_out/s5hip/*/*kernel_trace.csv")[0]
ks = [r for r in csv.DictReader(open(kf)) if int(r["Start_Timestamp"]) >= m]
per = collections.defaultdict(float)
for r in ks: per[r.get("Queue_Id")] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9
print("kernel seconds per queue:", dict(per))
# union of busy intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in ks)
busy = 0; cs, ce = iv[0]
for a, b in iv[1:]:
    if a > ce: busy += ce - cs; cs, ce = a, b
    else: ce = max(ce, b)
busy += ce - cs
print("device busy (union) %.3f s" % (busy / 1e9))
PY
rm -rf gpurun_out/s5hip
