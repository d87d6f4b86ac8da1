"""Where the pack + H2D time of a bench step goes (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from catch_amd import engine, probe
from catch_amd.utils import synthetic
groups = synthetic.dataset("S4")
ctx = engine.default_context()
tot = dict(concat=0.0, targets=0.0, cands=0.0, probes=0.0)
for rep in range(2):
    tot = dict(concat=0.0, targets=0.0, cands=0.0, probes=0.0)
    for genomes in groups:
        t0 = time.perf_counter()
        seqs = [s for g in genomes for s in g]
        buf, off = engine._concat(seqs)
        t1 = time.perf_counter()
        t = engine.Targets(ctx, genomes); ctx.sync()
        t2 = time.perf_counter()
        c = engine.Candidates(ctx, t, 100, 50); ctx.sync()
        t3 = time.perf_counter()
        k, ep, eo = probe.anchor_entries_equal_length(c.n, 100, 2, 100)
        p = c.probes(k, ep, eo); ctx.sync()
        t4 = time.perf_counter()
        tot["concat"] += t1 - t0; tot["targets"] += t2 - t1; tot["cands"] += t3 - t2; tot["probes"] += t4 - t3
        p.close(); c.close(); t.close()
    print({k: round(v, 3) for k, v in tot.items()}, "(targets includes its own concat)")
