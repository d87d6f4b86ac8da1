#!/bin/bash
# Run on the GPU box (through gpurun): collects the evidence that goes under
# profiles/ for the current round: the bench JSON line, the rocprofv3
# kernel-trace summary of the same command, and the HBM traffic counters
# (FETCH_SIZE / WRITE_SIZE in separate passes, as MI355X_MICROARCH.md says).
#   tools/collect_profiles.sh <tag> [workload] [extra]   -> gpurun_out/profiles_<tag>/
# extra = "all" also profiles the clustering pre-step and the device front end
tag=${1:-r04}; wl=${2:-S4}; extra=${3:-}
out=/root/repo/gpurun_out/profiles_$tag
rm -rf $out
mkdir -p $out
cd /root/repo
python bench.py --workload $wl > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python /root/repo/bench.py --workload $wl --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure --no-property-checks > $out/bench_under_rocprof.json 2>/dev/null
# the same with one chain at a time (the union instance after the large groups, not beside them): the summary whose
# per-kernel times back the line's `roofline` (a unit's launches with the device to themselves)
CATCHHIP_BENCH_UNION_BESIDE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_one_chain -- python /root/repo/bench.py --workload $wl --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure --no-property-checks --no-also > $out/bench_under_rocprof_one_chain.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python /root/repo/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure --no-property-checks > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python /root/repo/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure --no-property-checks > /dev/null 2>&1
if [ "$extra" = "all" ]; then
  # the kernels outside the bench's hot path: clustering pre-step and the device front end
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_cluster -- python /root/repo/tools/cluster_bench.py --scale 0.25 --cpu-sample 0 > $out/cluster_bench.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_frontend -- python /root/repo/tools/e2e_profile.py S4 0.5 > $out/frontend_e2e.txt 2>&1
fi
cd /root/repo
for t in cluster frontend; do
  f=$(ls -t $out/trace_$t/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $out/${t}_kernel_stats.csv
done
OUT=$out WL=$wl python - <<'PY'
import csv, glob, json, collections, shutil, os
out, wl = os.environ["OUT"], os.environ["WL"]
ks = max(glob.glob(out + "/trace/*/*kernel_stats.csv"), key=os.path.getmtime)
ds = sorted(glob.glob(out + "/trace/*/*domain_stats.csv"), key=os.path.getmtime)[-1:]
shutil.copy(ks, out + "/bench_kernel_stats.csv")
oc = glob.glob(out + "/trace_one_chain/*/*kernel_stats.csv")
if oc: shutil.copy(max(oc, key=os.path.getmtime), out + "/bench_kernel_stats_one_chain.csv")
if ds: shutil.copy(ds[0], out + "/bench_domain_stats.csv")
def pmc(dirname, counter):
    f = sorted(glob.glob(out + "/" + dirname + "/*/*counter_collection.csv"), key=os.path.getmtime)[-1:]
    acc = collections.defaultdict(lambda: [0.0, 0])
    if not f: return acc
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != counter: continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[name]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc
fe, wr = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
kern = {}
for k in sorted(set(fe) | set(wr)):
    kern[k] = {"FETCH_SIZE_KB_per_launch": fe[k][0] / max(fe[k][1], 1), "launches": fe[k][1] or wr[k][1],
               "WRITE_SIZE_KB_per_launch": wr[k][0] / max(wr[k][1], 1)}
def unit(names, per):
    """total bytes of the unit's kernels over the run / launches of the kernels in `per`
    (a solver round = one launch of each of its kernels: per = the count kernels)"""
    names = [n for n in names if n in kern]
    per = [n for n in per if n in kern]
    launches = sum(kern[n]["launches"] for n in per) or 1
    return {"kernels": names, "per_launch_of": per, "launches": launches,
            "FETCH_SIZE_KB_per_launch": sum(kern[n]["FETCH_SIZE_KB_per_launch"] * kern[n]["launches"] for n in names) / launches,
            "WRITE_SIZE_KB_per_launch": sum(kern[n]["WRITE_SIZE_KB_per_launch"] * kern[n]["launches"] for n in names) / launches}
k1a = [k for k in kern if k.startswith(("seed_init", "seed_count", "seed_alloc", "seed_fill", "seed_lookup", "kj_hitpos", "kj_compact",
                                        "radix_hist", "radix_scatter", "scan_tile_"))]
verify = [k for k in kern if k.startswith(("kj_verify", "kj_giant", "seed_verify"))]
rows = [k for k in kern if k.startswith(("scan1_", "bucket_", "rows_emit", "scan_tiles", "kj_bucket_count", "kj_bases"))]
setup = [k for k in kern if k.startswith(("gr_tile_", "set_ptr", "gr_bitmap"))]
solver = [k for k in kern if k.startswith(("gf_count_claim", "gf_check_apply", "gr_count", "gr_claim", "gr_check", "gr_apply", "gr_cover"))]
ndfk = [k for k in kern if k.startswith(("ndf_", "mh_"))]
rec = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of "
               "'python bench.py --workload %s --steps 2 --warmup 1 --no-cpu-baseline --no-property-checks' (3 steps, the two "
               "untimed steps that run one chain at a time and the untimed step with the E_dirty statistics = 6 steps in the run); KB per launch averaged over "
               "the launches of the run (all groups, all rounds); a unit = total KB of its kernels / launches of its "
               "leading kernel(s) (a solver round = one count launch with its claim / apply launches); on "
               "gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md), bench.py doubles it; "
               "other widths and WRITE_SIZE are uncalibrated" % wl,
       "workload": wl, "steps_in_run": 6,
       "disjoint_units": ["k1_table_hitpos_sort", "join_verify", "rows_build", "solver_setup", "solver_round", "ndf"],
       "units": {"solver_round": unit(solver, [k for k in solver if k.startswith(("gf_count_claim", "gr_count"))]),
                 "gr_claim": unit([k for k in solver if k.startswith("gr_claim")], [k for k in solver if k.startswith("gr_claim")]),
                 "join_verify": unit(verify, [k for k in verify if k.startswith(("kj_verify", "seed_verify"))]),
                 "k1_table_hitpos_sort": unit(k1a, [k for k in k1a if k.startswith(("kj_hitpos", "seed_lookup"))]),
                 "rows_build": unit(rows, [k for k in rows if k.startswith("rows_emit")]),
                 "solver_setup": unit(setup, [k for k in setup if k.startswith("gr_tile_scatter")]),
                 "ndf": unit(ndfk, [k for k in ndfk if k.startswith("ndf_key")][:1] or ndfk[:1])},
       "kernels": kern}
json.dump(rec, open(out + "/pmc_traffic.json", "w"), indent=1)
for i, r in enumerate(csv.DictReader(open(ks))):
    if i < 16: print(r["Name"][:44].ljust(44), r["Calls"].rjust(6), r["TotalDurationNs"].rjust(12), r["AverageNs"][:10].rjust(11))
print(open(out + "/bench.json").read()[:2500])
print(json.dumps(rec["units"], indent=0)[:1200])
PY
