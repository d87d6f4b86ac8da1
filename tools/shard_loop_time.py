"""GPU box: wall time of the universe-sharded round loop with ONE shard in this process (the whole group: the
kernels do what the unsharded solver's do) against the unsharded solve of the same rows -- what the loop's
per-round host work and read-backs cost.   python tools/shard_loop_time.py [group]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catch_amd import engine, parallel, probe
from catch_amd.utils import synthetic

gi = int(sys.argv[1]) if len(sys.argv) > 1 else 0
genomes = synthetic.dataset("S4")[gi]
ctx = engine.default_context()
full = engine.Targets(ctx, genomes)
cands = engine.Candidates(ctx, full, 100, 50)
k, ep, eo = probe.anchor_entries_equal_length(cands.n, 100, 2, 100)
probes = cands.probes(k, ep, eo)
rows = engine.Rows.scan(ctx, probes, full, 2, 100, 0, 50, 0)
for rep in range(3):
    ctx.sync()
    t0 = time.perf_counter()
    ids = rows.greedy(cands.n) if hasattr(rows, "greedy") else engine.setcover_greedy(ctx, rows, cands.n)
    ctx.sync()
    t1 = time.perf_counter()
    sh = engine.Shard(rows, cands.n)
    ctx.sync()
    t2 = time.perf_counter()
    rounds = [0]

    def exchange(which):
        rounds[0] += 1
        engine.shards_allreduce_local([sh], which)
    picks = parallel.sharded_solve([sh], exchange)
    ctx.sync()
    t3 = time.perf_counter()
    print("group %d: unsharded solve %.1f ms; shard set-up %.1f ms, sharded loop %.1f ms in %d rounds (%.2f ms per round); picks equal: %s"
          % (gi, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, rounds[0] // 2, (t3 - t2) * 1e3 / max(1, rounds[0] // 2),
             list(picks) == list(ids)))
    sh.close()
