"""GPU box: element counts of the packed exchange of a universe-sharded solve,
round by round (one S4 group split into `nshards` ranges on one device).
    python tools/shard_exchange_sizes.py [group] [nshards]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catch_amd import engine, parallel, probe
from catch_amd.utils import synthetic

gi = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 2
genomes = synthetic.dataset("S4")[gi]
ctx = engine.default_context()
full = engine.Targets(ctx, genomes)
cands = engine.Candidates(ctx, full, 100, 50)
k, ep, eo = probe.anchor_entries_equal_length(cands.n, 100, 2, 100)
probes = cands.probes(k, ep, eo)
b = parallel.split_universes([sum(len(s) for s in g) for g in genomes], ns)
shards = []
for r in range(ns):
    t = engine.Targets(ctx, genomes[b[r]:b[r + 1]])
    rows = engine.Rows.scan(ctx, probes, t, 2, 100, 0, 50, 0)
    shards.append(engine.Shard(rows, cands.n))
sizes = []
rnd = 0
while True:
    for sh in shards:
        sh.count()
    g = shards[0]._exchange_shape()[0]
    engine.shards_allreduce_local(shards, 0)
    for sh in shards:
        sh.claim_check()
    l = shards[0]._exchange_shape()[1]
    engine.shards_allreduce_local(shards, 1)
    done = [sh.apply() for sh in shards]
    sizes.append((rnd, g, l))
    rnd += 1
    if done[0]:
        break
picks = shards[0].picks()
print("sets %d, picks %d, rounds %d" % (cands.n, len(picks), rnd))
for r, g, l in sizes:
    print("round %2d: gains %9d x 4 B, marks %9d x 1 B" % (r, g, l))
tg, tl = sum(g for _, g, _ in sizes), sum(l for _, _, l in sizes)
print("packed: %.1f MB of gains + %.1f MB of marks; by set id: %.1f MB + %.1f MB"
      % (4e-6 * tg, 1e-6 * tl, 4e-6 * rnd * (cands.n + 2), 4e-6 * rnd * cands.n))
