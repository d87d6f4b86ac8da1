"""cProfile of one design_large step on S5 x scale (GPU box): where the host time of the clustering
and of the union filter goes."""
import cProfile, io, os, pstats, random, sys, time
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    print("cluster %.2f s, filters %.2f s (%s), %d clusters" % (t1 - t0, t2 - t1, mode, len(clusters)))
    print("stages (thread seconds):", {k: round(v, 2) for k, v in scf.last_timings.items() if k.endswith("_s")})
    from catch_amd.utils import cluster as _c
    print("components search:", _c._path_counts, _c._native_stats)
    print("clustering stages:", {k: round(v, 3) for k, v in getattr(pd, "cluster_timings", {}).items()})


if len(sys.argv) < 3 or sys.argv[2] != "once":
    step()
print("MARK monotonic_ns %d boottime_ns %d" % (time.clock_gettime_ns(time.CLOCK_MONOTONIC), time.clock_gettime_ns(time.CLOCK_BOOTTIME)), flush=True)
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60)
print(s.getvalue()[:14000])
