#!/usr/bin/env python3
"""Timing of the clustering pre-step (catch_amd/csrc/cluster.hip) on a synthetic
input: signatures, one distance row, the condensed matrix, and the whole
cluster_with_minhash_signatures for both methods.

    python tools/cluster_bench.py [--workload S4] [--scale 0.25] [--fragment 50000]
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from catch_amd import engine                      # noqa: E402
from catch_amd.genome import Genome               # noqa: E402
from catch_amd.utils import cluster, lsh, synthetic   # noqa: E402

PHASE_NDF = 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="S4")
    ap.add_argument("--scale", type=float, default=0.25)
    ap.add_argument("--fragment", type=int, default=50000)
    ap.add_argument("--cpu-sample", type=int, default=200000,
                    help="bases hashed by a hashlib loop for comparison (0 = skip)")
    ap.add_argument("--hierarchical-max", type=int, default=12000)
    a = ap.parse_args()
    groups = synthetic.dataset(a.workload, scale=a.scale)
    seqs = []
    for grp in groups:
        for g in grp:
            for s in g:
                seqs += Genome.from_one_seq(s).break_into_fragments(
                    a.fragment, include_full_end=True).seqs
    total = sum(len(s) for s in seqs)
    ctx = engine.default_context()
    out = dict(workload=a.workload, scale=a.scale, sequences=len(seqs), bases=total)
    fam = lsh.MinHashFamily(12, N=100)
    random.seed(1)
    fam.signatures(seqs[:4]).close()               # warm up (module load, pool)
    t0 = time.perf_counter()
    sigs = fam.signatures(seqs)
    out["signatures_wall_s"] = time.perf_counter() - t0
    ms, n = ctx.kernel_ms(PHASE_NDF)
    out["signatures_kernel_ms"] = ms
    # algorithmic bytes: 1 B/base read + 4 B/k-mer hash written, then the select
    # reads every hash 4 times (3 histogram levels + gather)
    out["signatures_GBps"] = total * (1 + 4 + 16) / (ms * 1e-3) / 1e9
    out["kmers_per_s"] = total / (ms * 1e-3)
    rows = []
    for j in range(0, len(seqs), max(1, len(seqs) // 20)):
        t0 = time.perf_counter()
        sigs.common_row(j)
        rows.append(time.perf_counter() - t0)
    out["row_wall_ms"] = 1e3 * float(np.median(rows))
    out["row_kernel_ms"] = ctx.kernel_ms(PHASE_NDF)[0]
    if len(seqs) <= a.hierarchical_max:
        lut = (1.0 - np.arange(101, dtype=np.float64) / 100.0).astype(np.float32)
        t0 = time.perf_counter()
        dm = sigs.condensed(lut)
        out["condensed_wall_s"] = time.perf_counter() - t0
        out["condensed_kernel_ms"] = ctx.kernel_ms(PHASE_NDF)[0]
        out["pairs"] = int(dm.size)
        out["pairs_per_s"] = dm.size / (out["condensed_kernel_ms"] * 1e-3)
    sigs.close()
    named = dict(enumerate(seqs))
    for method in ("simple", "hierarchical"):
        if method == "hierarchical" and len(seqs) > a.hierarchical_max:
            continue
        random.seed(1)
        t0 = time.perf_counter()
        cl = cluster.cluster_with_minhash_signatures(named, threshold=0.15, cluster_method=method)
        out[method + "_wall_s"] = time.perf_counter() - t0
        out[method + "_clusters"] = len(cl)
        out[method + "_largest"] = [len(c) for c in cl[:5]]
    if a.cpu_sample:
        import hashlib
        import heapq
        sample, got = [], 0
        for s in seqs:
            sample.append(s)
            got += len(s)
            if got >= a.cpu_sample:
                break
        t0 = time.perf_counter()
        for s in sample:      # the same arithmetic, one core, stdlib only
            heapq.nsmallest(100, ((12345 * int(hashlib.md5(s[i:i + 12].encode()).hexdigest(), 16) + 678)
                                  % (2 ** 31 - 1) for i in range(len(s) - 11)))
        dt = time.perf_counter() - t0
        out["cpu_signature_kmers_per_s"] = got / dt
        out["cpu_sample_bases"] = got
    print(json.dumps(out))


if __name__ == "__main__":
    main()
