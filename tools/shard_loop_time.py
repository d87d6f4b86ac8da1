"""Round loop of a sharded solve, timed (GPU box): S4's largest group cut into N universe ranges on ONE device, the
interpreter's loop (parallel.sharded_solve: one library call per step, one read-back per round, exact exchange sizes)
against catchhip_shard_solve with 1 / 2 / 4 / 8 rounds per read-back.   tools/shard_loop_time.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")
from catch_amd import engine, parallel, probe
from catch_amd.utils import synthetic

ns = [int(x) for x in sys.argv[1:]] or [1, 4, 8]
genomes = synthetic.dataset("S4")[0]
ctx = engine.default_context()
full = engine.Targets(ctx, genomes)
cands = engine.Candidates(ctx, full, 100, 50)
k, ep, eo = probe.anchor_entries_equal_length(cands.n, 100, 2, 100)
probes = cands.probes(k, ep, eo)
for n in ns:
    b = parallel.split_universes([sum(len(s) for s in g) for g in genomes], n)
    held = []
    for r in range(n):
        t = engine.Targets(ctx, genomes[b[r]:b[r + 1]])
        held += [t, engine.Rows.scan(ctx, probes, t, 2, 100, 0, 50, 0)]
    rows = held[1::2]
    ref = None
    for label, rps in (("python loop", None), ("C loop, 1 round / read-back", 1), ("C loop, 2", 2), ("C loop, 4", 4), ("C loop, 8", 8)):
        best = None
        for _ in range(2):
            shards = [engine.Shard(rw, cands.n) for rw in rows]
            ctx.sync()
            t0 = time.perf_counter()
            if rps is None:
                picks = parallel.sharded_solve(shards, lambda w: engine.shards_allreduce_local(shards, w))
            else:
                picks = engine.shards_solve(shards, "local", rps)
            ctx.sync()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
            for sh in shards:
                sh.close()
        ref = picks if ref is None else ref
        print("N = %d  %-30s %8.2f ms for all shards (%.2f per shard)  picks %d %s" % (n, label, best, best / n, len(picks), "ok" if picks == ref else "DIFFERENT"))
    for h in held[::-1]:
        h.close()
