#!/usr/bin/env python3
"""MinHash near-duplicate filter over many clusters in one pass: wall time and
device time.   python tools/ndf_many_bench.py [scale]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catch_amd import engine                                             # noqa: E402
from catch_amd.filter import candidate_probes                            # noqa: E402
from catch_amd.filter.near_duplicate_filter import NearDuplicateFilterWithMinHash  # noqa: E402
from catch_amd.utils import synthetic                                    # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
    genomes = synthetic.dataset("S5", scale=scale)[0]
    groups = [candidate_probes.candidate_strings_from_sequences(list(g), 100, 50) for g in genomes]
    f = NearDuplicateFilterWithMinHash(0.6)
    ctx = engine.default_context()
    for rep in range(2):
        random.seed(3)
        t0 = time.perf_counter()
        out = f._filter_strs_many(groups)
        dt = time.perf_counter() - t0
        ms, nl = ctx.kernel_ms(engine.PHASE_NDF)
        print("groups %d probes %d -> %d kept; wall %.3f s; device phase %.1f ms in %d launches"
              % (len(groups), sum(map(len, groups)), sum(map(len, out)), dt, ms, nl), flush=True)


if __name__ == "__main__":
    main()
