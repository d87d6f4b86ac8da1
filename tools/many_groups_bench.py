#!/usr/bin/env python3
"""Set cover filter over many small groups (the clusters of a config-5 design):
wall time per group as a function of CATCHHIP_GROUPS_IN_FLIGHT.
    python tools/many_groups_bench.py [scale]"""
import os
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")   # (the tools below switch code paths through test hooks)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catch_amd.filter import candidate_probes                    # noqa: E402
from catch_amd.filter.set_cover_filter import SetCoverFilter    # noqa: E402
from catch_amd.genome import Genome                              # noqa: E402
from catch_amd.utils import synthetic                            # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
    genomes = synthetic.dataset("S5", scale=scale)[0]
    # one group per species-like genome (what clustering produces at this scale)
    groups = [[Genome.from_one_seq(g[0])] for g in genomes][:1200]
    cands = [list(dict.fromkeys(candidate_probes.candidate_strings_from_sequences(
        list(g[0].seqs), 100, 50))) for g in groups]
    f = SetCoverFilter(mismatches=5, lcf_thres=100, coverage=1.0, cover_extension=50)
    for inflight in (1, 2, 4, 8, 16):
        os.environ["CATCHHIP_GROUPS_IN_FLIGHT"] = str(inflight)
        best = None
        for _ in range(2):
            np.random.seed(1)
            t0 = time.perf_counter()
            out = f._filter_strs(cands, groups, assume_unique=True)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print("in flight %2d: %.3f s for %d groups = %.3f ms per group (%d probes picked)"
              % (inflight, best, len(groups), 1e3 * best / len(groups), sum(map(len, out))), flush=True)


if __name__ == "__main__":
    main()
