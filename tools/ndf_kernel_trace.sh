#!/bin/bash
# usage (GPU box): tools/ndf_kernel_trace.sh   -> kernel totals of one S5 x 1.0 design_large step with ONE front-end worker (kernels with the
# device to themselves) and, per union chunk, the near-duplicate filter's launches: round-0 pass, probe passes, drain, wake-ups
export CATCHHIP_TEST_HOOKS=1 CATCHHIP_FRONT_END_WORKERS=1 CATCHHIP_PREFETCH_DEPTH=0
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s5kt -- python $R/tools/s5_profile.py 1.0 once > $R/gpurun_out/s5kt.out 2>&1
cd $R
f=$(ls gpurun_out/s5kt/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/s5_kernel_stats_q.csv
python tools/kstats.py $f 14
# per-dispatch: lazy kernels of the first chunk
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/s5kt/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = 0
out = []
for r in rows:
    nm = r["Kernel_Name"]
    if "ndf_lazy" in nm or "ndf_wake" in nm or "ndf_probe" in nm:
        tag = "probe" if "ndf_probe" in nm else "drain" if "Lb1ELb1E" in nm or "true, true" in nm else ("wake" if "wake" in nm else "pass")
        out.append((tag, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size") or r.get("Grid_Size_X")))
chunks = []
for o in out:
    if o[0] == "pass":
        chunks.append(dict(r0_pass=o[1], r0_drain=0.0, probe=0.0, drain=0.0, wake=0.0, rounds=0, first=True))
        continue
    if not chunks: continue
    c = chunks[-1]
    if o[0] == "drain":
        if c["first"]: c["r0_drain"] = o[1]; c["first"] = False
        else: c["drain"] += o[1]
    elif o[0] == "probe": c["probe"] += o[1]; c["rounds"] += 1
    elif o[0] == "wake": c["wake"] += o[1]
tot = {}
for c in chunks:
    c.pop("first")
    for k, v in c.items(): tot[k] = tot.get(k, 0) + v
    print({k: round(v / 1e3, 2) if k != "rounds" else v for k, v in c.items()})
print("TOTAL ms", {k: round(v / 1e3, 1) if k != "rounds" else v for k, v in tot.items()})
PY
rm -rf gpurun_out/s5kt
