"""One rank's scan of a sharded group (GPU box, under rocprofv3 --kernel-trace --stats): all candidates of S4's largest
group against 1/N of its genomes, three times.   tools/shard_scan_profile.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")
from catch_amd import engine, parallel, probe
from catch_amd.utils import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
genomes = synthetic.dataset("S4")[0]
ctx = engine.default_context()
full = engine.Targets(ctx, genomes)
cands = engine.Candidates(ctx, full, 100, 50)
k, ep, eo = probe.anchor_entries_equal_length(cands.n, 100, 2, 100)
probes = cands.probes(k, ep, eo)
b = parallel.split_universes([sum(len(s) for s in g) for g in genomes], n)
t = engine.Targets(ctx, genomes[b[0]:b[1]])
for _ in range(3):
    ctx.sync()
    t0 = time.perf_counter()
    rows = engine.Rows.scan(ctx, probes, t, 2, 100, 0, 50, 0)
    ctx.sync()
    print("scan of 1/%d: %.2f ms, %d rows" % (n, (time.perf_counter() - t0) * 1e3, rows.n), file=sys.stderr)
    rows.close()
