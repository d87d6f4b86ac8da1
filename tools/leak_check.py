#!/usr/bin/env python3
"""40 designs through every path (device front end per group and as one grouped
instance, all three first filters, adapters, clustering, coverage analysis) and
the device memory in use after 5, 10, 20, 30 and 40 of them (rocm-smi): the
caching allocator must reach a steady state.   python tools/leak_check.py"""
import os, sys, subprocess, re, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from catch_amd.filter import duplicate_filter, near_duplicate_filter, probe_designer, set_cover_filter
from catch_amd.filter.adapter_filter import AdapterFilter
from catch_amd.genome import Genome
from catch_amd.utils import synthetic
from catch_amd import coverage_analysis
def used():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout
    m = re.search(r"Used Memory \(B\): (\d+)", out)
    return int(m.group(1)) if m else -1
rng = np.random.Generator(np.random.PCG64(1))
groups = [[Genome.from_one_seq(g[0]) for g in synthetic.make_species(rng, [6000], 12, 3, 0.05, 0.01)] for _ in range(3)]
marks = []
for it in range(40):
    kind = it % 3
    first = (duplicate_filter.DuplicateFilter() if kind == 0 else near_duplicate_filter.NearDuplicateFilterWithHammingDistance(2, 100)
             if kind == 1 else near_duplicate_filter.NearDuplicateFilterWithMinHash(0.6))
    scf = set_cover_filter.SetCoverFilter(mismatches=2 + (it % 2) * 3, lcf_thres=100, cover_extension=50)
    af = AdapterFilter(("AA", "CC"), ("GG", "TT"), 2, 100)
    pd = probe_designer.ProbeDesigner(groups, [first, scf, af], probe_length=100, probe_stride=50,
                                      cluster_threshold=0.15 if it % 4 == 0 else None, cluster_merge_after=scf if it % 4 == 0 else None,
                                      cluster_method="choose" if it % 4 == 0 else None, cluster_fragment_length=2000 if it % 4 == 0 else None)
    random.seed(it); np.random.seed(it)
    pd.design()
    if it % 5 == 0:
        an = coverage_analysis.Analyzer(pd.final_probes[:50], 2, 100, groups, rc_too=True)
        an.run()
    if it in (4, 9, 19, 29, 39):
        marks.append((it, used()))
print(marks)
print("growth after warm-up (MB):", (marks[-1][1] - marks[1][1]) / 1e6)
