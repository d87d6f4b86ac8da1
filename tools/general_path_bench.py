#!/usr/bin/env python3
"""Scan time of the general path (cover threshold below the probe length, or an
island of exact match) next to the seed scan, same probes and targets.
    python tools/general_path_bench.py [workload] [scale]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catch_amd import engine, probe                    # noqa: E402
from catch_amd.filter import candidate_probes          # noqa: E402
from catch_amd.utils import synthetic                  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "S2"
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    genomes = synthetic.dataset(wl, scale=scale)[0]
    strs = list(dict.fromkeys(s for g in genomes for s in
                              candidate_probes.candidate_strings_from_sequences(list(g), 100, 50)))
    ctx = engine.default_context()
    t = engine.Targets(ctx, genomes)
    print("targets %d bp, %d candidates" % (t.total, len(strs)))
    for name, m, thres, island in (("seed scan  -m 2 -l 100", 2, 100, 0), ("general    -m 2 -l 80", 2, 80, 0),
                                   ("general    -m 2 -l 100 island 30", 2, 100, 30),
                                   ("seed scan  -m 5 -l 100 (random anchors)", 5, 100, 0),
                                   ("general    -m 5 -l 80", 5, 80, 0)):
        np.random.seed(1)
        k, uniq, owner, ep, eo = probe.anchor_table(strs, m, thres, assume_unique=True)
        p = engine.Probes(ctx, uniq, owner, ep, eo, k)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            rows = engine.Rows.scan(ctx, p, t, m, thres, island, 50)
            dt = time.perf_counter() - t0
            n = rows.n
            rows.close()
            best = dt if best is None else min(best, dt)
        ms_scan = ctx.kernel_ms(engine.PHASE_SCAN)[0]
        ms_rows = ctx.kernel_ms(engine.PHASE_ROWS)[0]
        print("%-42s wall %.2f ms  scan %.2f ms  rows %.2f ms  (%d rows, %d anchors)"
              % (name, 1e3 * best, ms_scan, ms_rows, n, len(ep)))
        p.close()
    t.close()


if __name__ == "__main__":
    main()
