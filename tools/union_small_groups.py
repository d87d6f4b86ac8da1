"""GPU box: the smaller S4 groups solved one after the other (resident objects, as bench.py's step does) against
the same groups as ONE instance (targets / candidates / probes that carry group numbers: one table, one scan,
one solve whose rounds cover all groups at once).   python tools/union_small_groups.py [max Mbases per group]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from catch_amd import engine, probe
from catch_amd.utils import synthetic

limit = float(sys.argv[1]) * 1e6 if len(sys.argv) > 1 else 16e6
groups = synthetic.dataset("S4")
bases = [sum(len(s) for g in grp for s in g) for grp in groups]
small = [i for i in range(len(groups)) if bases[i] < limit]
print("groups below %.0f Mbases: %s (%.1f Mbases in all)" % (limit / 1e6, small, sum(bases[i] for i in small) / 1e6))
ctx = engine.default_context()
res = []
for gi in small:
    t = engine.Targets(ctx, groups[gi])
    c = engine.Candidates(ctx, t, 100, 50)
    k, ep, eo = probe.anchor_entries_equal_length(c.n, 100, 2, 100)
    res.append((t, c, c.probes(k, ep, eo)))


def per_group():
    out = []
    for t, c, p in res:
        ids, nrows = engine.setcover_filter(ctx, p, t, 2, 100, 0, 50, c.n)
        out.append(list(ids))
    return out


genomes = [g for gi in small for g in groups[gi]]
ngen = [len(groups[gi]) for gi in small]
ut = engine.Targets(ctx, genomes)
ut.set_groups(np.repeat(np.arange(len(small)), ngen))
uc = engine.Candidates(ctx, ut, 100, 50)
k, ep, eo = probe.anchor_entries_equal_length(uc.n, 100, 2, 100)
up = uc.probes(k, ep, eo)
first = np.concatenate([[0], np.cumsum(np.bincount(uc.groups(), minlength=len(small)))])


def union():
    ids, nrows = engine.setcover_filter(ctx, up, ut, 2, 100, 0, 50, uc.n)
    ids = np.asarray(ids, dtype=np.int64)
    grp = uc.groups()[ids]
    return [(ids[grp == g] - first[g]).tolist() for g in range(len(small))]


for name, fn in (("one after the other", per_group), ("one instance", union)):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        got = fn()
    ctx.sync()
    print("%-20s %.2f ms per pass" % (name, (time.perf_counter() - t0) / 3 * 1e3))
    if name == "one after the other":
        want = got
print("same picks per group, in each group's own order:", got == want)
