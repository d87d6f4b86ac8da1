"""Rows of an S4 group by the level (log2) of their set's INITIAL gain: what a level-by-level schedule of the solver's
rounds would leave out of the first rounds (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")
import numpy as np
from catch_amd import engine, probe
from catch_amd.utils import synthetic
gi = int(sys.argv[1]) if len(sys.argv) > 1 else 0
groups = synthetic.dataset("S4")
genomes = groups[gi]
ctx = engine.default_context()
t = engine.Targets(ctx, genomes)
c = engine.Candidates(ctx, t, 100, 50)
k, ep, eo = probe.anchor_entries_equal_length(c.n, 100, 2, 100)
p = c.probes(k, ep, eo)
rows = engine.Rows.scan(ctx, p, t, 2, 100, 0, 50, 0)
sid, un, st, en = rows.fetch()
ln = (en - st).astype(np.int64)
gain = np.bincount(sid, weights=ln, minlength=c.n).astype(np.int64)
nrows = np.bincount(sid, minlength=c.n)
lvl = np.zeros(c.n, dtype=np.int64)
nz = gain > 0
lvl[nz] = np.floor(np.log2(gain[nz])).astype(np.int64)
print("sets %d rows %d max gain %d" % (c.n, sid.size, gain.max()))
tot = sid.size
cum = 0
for L in range(int(lvl.max()), -1, -1):
    m = lvl == L
    r = int(nrows[m].sum())
    cum += r
    if r:
        print("level %2d (gain >= %7d): %8d sets %10d rows (%5.1f %%), cumulative %5.1f %%" % (L, 1 << L, int(m.sum()), r, 100.0 * r / tot, 100.0 * cum / tot))
