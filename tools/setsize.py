import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from catch_amd import engine, probe
from catch_amd.filter import candidate_probes
from catch_amd.utils import synthetic
groups = synthetic.dataset("S4")
ctx = engine.default_context()
for gi in (0, 5, 12):
    genomes = groups[gi]
    seqs = [s for g in genomes for s in g]
    c = list(dict.fromkeys(candidate_probes.candidate_strings_from_sequences(seqs, 100, 50)))
    k, uniq, owner, ep, eo = probe.anchor_table(c, 2, 100, assume_unique=True)
    t = engine.Targets(ctx, genomes); p = engine.Probes(ctx, uniq, owner, ep, eo, k)
    rows = engine.Rows.scan(ctx, p, t, 2, 100, 0, 50)
    sid, un, st, en = rows.fetch()
    cnt = np.bincount(sid, minlength=len(c))
    print("group", gi, "genomes", len(genomes), "sets", len(c), "rows", rows.n, "max rows/set", cnt.max(),
          "pct", np.percentile(cnt, [50, 90, 99, 99.9]).tolist(), "sets>1024:", int((cnt > 1024).sum()))
    rows.close(); p.close(); t.close()
