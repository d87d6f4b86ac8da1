#!/usr/bin/env python3
"""Throughput of SURVEY row a13 (SetCoverFilter._make_ranks under --identify: every group's candidates scanned, with
the tolerant model, against ALL groups' genomes and their reverse complements; catch/filter/set_cover_filter.py:472-529,
614-735) on S4 x scale (GPU box).   python tools/ranks_bench.py [scale] > profiles/r04_ranks_bench.json"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")
from catch_amd import engine, genome  # noqa: E402
from catch_amd.filter import candidate_probes  # noqa: E402
from catch_amd.filter.set_cover_filter import SetCoverFilter  # noqa: E402
from catch_amd.utils import synthetic  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
groups = synthetic.dataset("S4", scale=scale)
gen = [[genome.Genome.from_one_seq(g[0]) for g in grp] for grp in groups]
cands = [list(dict.fromkeys(candidate_probes.candidate_strings_from_sequences([s for g in grp for s in g], 100, 50)))
         for grp in groups]
G_all = sum(len(s) for grp in groups for g in grp for s in g)
f = SetCoverFilter(mismatches=2, lcf_thres=100, coverage=1.0, cover_extension=50, identify=True,
                   mismatches_tolerant=3, lcf_thres_tolerant=100)
ctx = engine.default_context()
res = []
for rep in range(2):
    ctx.sync()
    t0 = time.perf_counter()
    nranks = []
    for gi in range(len(groups)):
        r = f._make_ranks_strs(cands[gi], gen, ctx)
        nranks.append(int(r.max()) + 1 if len(r) else 0)
    ctx.sync()
    res.append(time.perf_counter() - t0)
units = sum(len(c) for c in cands) * 2.0 * G_all           # every candidate against every genome and its reverse complement
json.dump({"workload": "S4 x %g: --identify ranks of all %d groups (tolerant model -mt 3 -lt 100; each group's candidates "
                       "against all groups' genomes + reverse complements)" % (scale, len(groups)),
           "candidates": sum(len(c) for c in cands), "target_bases": G_all, "seconds": min(res), "seconds_all": res,
           "value": units / min(res), "unit": "probe*bp/s (incl. reverse complements)", "distinct_ranks_per_group": nranks,
           "note": "host strings in (reverse complements made on the host, targets packed per call), dense ranks out"},
          sys.stdout, indent=1)
