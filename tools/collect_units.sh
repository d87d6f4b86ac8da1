#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 evidence of one bench workload by UNIT of the hot path -- the kernel
# summary (--kernel-trace --stats) and three PMC passes of the same command (FETCH_SIZE; WRITE_SIZE; SQ_INSTS_VALU with
# GRBM_GUI_ACTIVE -- each in a run of its own, as MI355X_MICROARCH.md prescribes).
#   tools/collect_units.sh <tag> <workload> [scale] [steps]   -> gpurun_out/units_<tag>_<workload>/
#       kernel_stats.csv, bench_under_rocprof.json, pmc.json (what bench.py's s5_roofline reads as profiles/<tag>_pmc_<workload>.json)
tag=${1:-r05}; wl=${2:-S5}; sc=${3:-1.0}; steps=${4:-1}
R=$PWD
out=$R/gpurun_out/units_${tag}_${wl}
rm -rf $out; mkdir -p $out
export CATCHHIP_TEST_HOOKS=1
args="--workload $wl --scale $sc --steps $steps --warmup 1 --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure --no-property-checks --no-solver-check --no-also"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py $args > $out/bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python $R/bench.py $args > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python $R/bench.py $args > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_valu -- python $R/bench.py $args > /dev/null 2>&1
cd $R
OUT=$out WL=$wl SC=$sc STEPS=$steps python - <<'PY'
import csv, glob, json, collections, shutil, os
out, wl, sc, steps = os.environ["OUT"], os.environ["WL"], float(os.environ["SC"]), int(os.environ["STEPS"]) + 1
ks = max(glob.glob(out + "/trace/*/*kernel_stats.csv"), key=os.path.getmtime)
shutil.copy(ks, out + "/kernel_stats.csv")
UNITS = [("ndf", ("ndf_", "mh_", "pyset_", "cand_pyhash")),
         ("clustering", ("sig_", "kmer_md5", "dfs_", "comp_")),
         ("front_end", ("cand_",)),
         ("k1_table_hitpos_sort", ("seed_init", "seed_count", "seed_alloc", "seed_fill", "seed_lookup", "kj_hitpos", "kj_compact",
                                   "scan_tile_", "seed_list")),
         ("join_verify", ("kj_verify", "kj_giant", "seed_verify")),
         ("rows_build", ("scan1_", "bucket_", "rows_emit", "scan_tiles", "kj_bucket_count", "kj_bases")),
         ("solver_setup", ("gr_tile_", "set_ptr", "gr_bitmap", "gf_build", "gf_universe")),
         ("solver_round", ("gf_count_claim", "gf_check_apply", "gr_count", "gr_claim", "gr_check", "gr_apply", "gr_usel", "gr_finish",
                           "gr_verdict", "gr_fixup", "gr_seg", "gr_cover")),
         ("radix_sort", ("radix_",))]
def unit_of(name):
    n = name.split("(")[0].replace("void ", "").strip()
    for u, pre in UNITS:
        if n.startswith(pre):
            return u
    return "other"
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))     # kernel -> counter -> total over the run
def pmc(dirname):
    f = sorted(glob.glob(out + "/" + dirname + "/*/*counter_collection.csv"), key=os.path.getmtime)[-1:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    if not f: return acc
    for r in csv.DictReader(open(f[0])):
        acc[unit_of(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        per_kernel[r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc
fe, wr, va = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_valu")
# device time and launches per unit from the kernel summary of the plain trace run
tms, calls, top = collections.defaultdict(float), collections.defaultdict(int), collections.defaultdict(list)
for r in csv.DictReader(open(ks)):
    u = unit_of(r["Name"])
    tms[u] += float(r["TotalDurationNs"]) / 1e6
    calls[u] += int(r["Calls"])
    top[u].append((float(r["TotalDurationNs"]) / 1e6, r["Name"].split("(")[0].replace("void ", "")[:60], int(r["Calls"])))
units = {}
for u in sorted(set(tms) | set(fe) | set(wr) | set(va)):
    gui = va[u].get("GRBM_GUI_ACTIVE", 0.0)
    valu = va[u].get("SQ_INSTS_VALU", 0.0)
    units[u] = {"kernel_ms_in_run": tms[u], "launches_in_run": calls[u],
                "FETCH_SIZE_KB": fe[u].get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB": wr[u].get("WRITE_SIZE", 0.0),
                "SQ_INSTS_VALU": valu, "SQ_WAVES": va[u].get("SQ_WAVES", 0.0), "GRBM_GUI_ACTIVE": gui,
                # a 64-wide wavefront occupies its 16-lane SIMD for 4 cycles per VALU instruction; 256 CUs x 4 SIMDs; the
                # busy cycles of the 8 XCDs are summed in GRBM_GUI_ACTIVE
                "valu_issue_frac": (valu * 4.0 / (1024.0 * gui / 8.0)) if gui else None,
                "top_kernels_ms": sorted(top[u], reverse=True)[:6]}
rec = {"note": "rocprofv3 passes of 'python bench.py --workload %s --scale %g --steps %d --warmup 1 --no-cpu-baseline ...' "
               "(%d steps in the run): kernel summary (plain --kernel-trace --stats run), --pmc FETCH_SIZE, --pmc WRITE_SIZE and "
               "--pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES in runs of their own; totals over the run by unit of the hot path "
               "(kernels by name prefix); on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md): "
               "bench.py prices traffic as 2 x FETCH_SIZE + WRITE_SIZE" % (wl, sc, steps - 1, steps),
       "workload": wl, "scale": sc, "steps_in_run": steps, "units": units,
       "kernels_by_fetch": [dict(kernel=k, FETCH_SIZE_KB=v.get("FETCH_SIZE", 0.0), WRITE_SIZE_KB=v.get("WRITE_SIZE", 0.0),
                                 SQ_INSTS_VALU=v.get("SQ_INSTS_VALU", 0.0))
                            for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0.0))[:24]]}
json.dump(rec, open(out + "/pmc.json", "w"), indent=1)
for u, v in sorted(units.items(), key=lambda kv: -kv[1]["kernel_ms_in_run"]):
    print("%-22s %9.1f ms %7d launches  fetch %8.2f GB  write %8.2f GB  valu %s" % (
        u, v["kernel_ms_in_run"], v["launches_in_run"], v["FETCH_SIZE_KB"] / 1e6, v["WRITE_SIZE_KB"] / 1e6,
        ("%.3f" % v["valu_issue_frac"]) if v["valu_issue_frac"] is not None else "-"))
for r in rec["kernels_by_fetch"][:14]:
    print("  %-56s fetch %8.1f GB  write %8.1f GB" % (r["kernel"], r["FETCH_SIZE_KB"] / 1e6, r["WRITE_SIZE_KB"] / 1e6))
print(open(out + "/bench_under_rocprof.json").read()[:300])
PY
rm -rf $out/trace $out/pmc_fetch $out/pmc_write $out/pmc_valu
