"""Wall time of the design_large step on S5 x scale per environment setting (GPU box):
   tools/s5_time.py 1.0 "" "CATCHHIP_FRONT_END_WORKERS=1" ...   -> best of 2 after a warm-up, cluster / filters split."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")
import numpy as np
from catch_amd import genome
from catch_amd.filter import near_duplicate_filter, probe_designer, set_cover_filter
from catch_amd.utils import synthetic

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
genomes = synthetic.dataset("S5", scale=scale)[0]
gobjs = [[genome.Genome.from_one_seq(g[0]) for g in genomes]]


def step():
    random.seed(21); np.random.seed(22)
    ndf = near_duplicate_filter.NearDuplicateFilterWithMinHash(0.6)
    scf = set_cover_filter.SetCoverFilter(mismatches=5, lcf_thres=100, coverage=1.0, cover_extension=50, kmer_probe_map_k=20)
    pd = probe_designer.ProbeDesigner(gobjs, [ndf, scf], probe_length=100, probe_stride=50, cluster_threshold=0.15,
                                      cluster_merge_after=scf, cluster_method="choose", cluster_fragment_length=50000)
    t0 = time.perf_counter()
    clusters = pd._cluster_genomes()
    t1 = time.perf_counter()
    mode = pd._device_front_end_mode(clusters, ndf, scf)
    run = scf._filter_genomes_device if mode == "per group" else scf._filter_genomes_device_union
    chosen = run(clusters, 100, 50, None, ndf)
    t2 = time.perf_counter()
    import hashlib
    h = hashlib.sha256()
    for c in chosen:
        for p in c:
            h.update(p.encode()); h.update(b"\n")
        h.update(b"|")
    if os.environ.get("S5_TIME_EVENTS"):
        for ev in sorted(scf.last_timings.get("pipe_events", []), key=lambda e: e[2]):
            print("   %-8s %3d  %.3f -> %.3f  (%.3f)" % (ev[0], ev[1], ev[2], ev[3], ev[3] - ev[2]))
    tm = {k: round(v, 2) for k, v in scf.last_timings.items() if k.endswith("_s")}
    tm["cluster"] = {k: round(v, 3) for k, v in getattr(pd, "cluster_timings", {}).items()}
    return t1 - t0, t2 - t1, h.hexdigest()[:12], tm


step()
for setting in (sys.argv[2:] or [""]):
    saved = {}
    for kv in setting.split():
        k, v = kv.split("=", 1)
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    best = None
    for _ in range(2):
        r = step()
        if best is None or r[0] + r[1] < best[0] + best[1]:
            best = r
    print("%-40s cluster %.2f s  filters %.2f s  total %.2f s  digest %s  %s" % (setting or "-", best[0], best[1], best[0] + best[1], best[2], best[3]))
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
