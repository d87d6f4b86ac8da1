#!/usr/bin/env python3
"""Digests of the selected probes at the FULL sizes of BASELINE configs[2..4].

Authoring container only (minutes to an hour of CPU).  The GPU box cannot run
the CPU oracle at these sizes inside a test, so the expected answers are
computed here -- with the oracle that tests/test_oracle_golden.py pins to the
reference (threaded per-sequence scans + the lazy evaluation of the greedy,
which that file proves equal to the line-by-line restatement) -- and committed
as data: per group (n_candidates, n_picks, sha256 of the sorted pick ids and of
the ids in pick order; the same for -c 0.9 on the same rows).
tests/test_gpu_parity.py::test_full_size_* assert them on the MI355X.

    python tests/golden/make_full_size.py S4 [S3] [S5:0.05]  -> tests/golden/full_size_picks.json
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

from catch_amd.filter import candidate_probes  # noqa: E402
from catch_amd.utils import synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402

OUT = os.environ.get("FULL_SIZE_OUT", os.path.join(HERE, "full_size_picks.json"))   # (FULL_SIZE_OUT: several jobs side by side)
L, STRIDE, MISMATCHES, EXT = 100, 50, 2, 50


def digest(ids):
    a = np.sort(np.asarray(ids, dtype=np.int64))
    return hashlib.sha256(a.astype("<i8").tobytes()).hexdigest()


def digest_in_order(ids):
    """Order-sensitive: the ids in the sequential pick order."""
    return hashlib.sha256(np.asarray(ids, dtype="<i8").tobytes()).hexdigest()


def set_cover_groups(name, scale):
    groups = synthetic.dataset(name, scale=scale)
    recs = []
    for gi, genomes in enumerate(groups):
        t0 = time.perf_counter()
        seqs = [s for g in genomes for s in g]
        cands = list(dict.fromkeys(
            candidate_probes.candidate_strings_from_sequences(seqs, L, STRIDE)))
        k, entries = orc.anchor_table(cands, MISMATCHES, L)
        rows = orc.make_sets(cands, entries, k, genomes, MISMATCHES, L, 0, EXT)
        glen = [sum(len(s) for s in g) for g in genomes]
        picks = orc.lazy_greedy(rows[0], rows[1], rows[2], rows[3], len(cands), glen)
        # the same rows under partial coverage (-c 0.9: 90 % of every genome)
        picks09 = orc.lazy_greedy(rows[0], rows[1], rows[2], rows[3], len(cands), glen,
                                  universe_p=[0.9] * len(genomes))
        recs.append(dict(group=gi, genomes=len(genomes),
                         bases=sum(len(s) for s in seqs),
                         n_candidates=len(cands), n_rows=int(rows[0].size),
                         n_picks=len(picks), picks_sha256=digest(picks),
                         picks_in_order_sha256=digest_in_order(picks),
                         n_picks_c09=len(picks09), picks_c09_sha256=digest(picks09),
                         picks_c09_in_order_sha256=digest_in_order(picks09)))
        sys.stderr.write("%s x%g group %d: %s (%.0f s)\n"
                         % (name, scale, gi, recs[-1], time.perf_counter() - t0))
        sys.stderr.flush()
        del rows
    return recs


def config3(scale):
    """configs[2]: S3 with --filter-with-lsh-hamming 2 in front of the set cover
    (bin/design.py:296-340): candidate windows of all records -> Hamming
    near-duplicate filter (random.seed(7) draws the sampled positions) ->
    SetCoverFilter over the one group."""
    import random
    groups = synthetic.dataset("S3", scale=scale)
    genomes = groups[0]
    t0 = time.perf_counter()
    seqs = [s for g in genomes for s in g]
    strs = candidate_probes.candidate_strings_from_sequences(seqs, L, STRIDE)
    random.seed(7)
    pos = orc.lsh_draw_positions(orc.lsh_num_tables(2, L, 20), 20, L)
    kept = orc.ndf_hamming_c(strs, 2, pos)
    t1 = time.perf_counter()
    k, entries = orc.anchor_table(kept, MISMATCHES, L)
    rows = orc.make_sets(kept, entries, k, genomes, MISMATCHES, L, 0, EXT)
    picks = orc.lazy_greedy(rows[0], rows[1], rows[2], rows[3], len(kept),
                            [sum(len(s) for s in g) for g in genomes])
    rec = dict(group=0, genomes=len(genomes), bases=sum(len(s) for s in seqs),
               n_windows=len(strs), n_candidates=len(kept),
               kept_sha256=hashlib.sha256("\n".join(kept).encode()).hexdigest(),
               n_rows=int(rows[0].size), n_picks=len(picks),
               picks_sha256=digest(picks), picks_in_order_sha256=digest_in_order(picks))
    sys.stderr.write("S3 x%g: %s (ndf %.0f s, total %.0f s)\n"
                     % (scale, rec, t1 - t0, time.perf_counter() - t0))
    return dict(flags="-pl 100 -ps 50 -m 2 -e 50 -c 1.0 --filter-with-lsh-hamming 2 "
                      "(random.seed(7) before the filter is built); pick ids index "
                      "the near-duplicate filter's output in its inclusion order",
                seed="synthetic.dataset default", groups=[rec])


def config5(scale):
    """configs[4]: the design_large chain on S5 x scale, restated with the
    oracle exactly as tests/test_gpu_parity.py::test_config5_pipeline_... does
    for its toy input: 50-kb fragments -> MinHash-signature clustering at 0.15
    -> per cluster candidates, MinHash near-duplicate filter 0.6, set cover
    with -m 5 -e 50 (random anchors: np.random.seed(22); random.seed(21))."""
    import random
    groups = synthetic.dataset("S5", scale=scale)
    genomes = groups[0]
    t0 = time.perf_counter()
    frags = [f for g in genomes for s in g for f in orc.fragments_of(s, 50000)]
    random.seed(21)
    np.random.seed(22)
    clusters = orc.cluster_with_minhash_signatures(frags, threshold=0.15,
                                                   cluster_method="simple")
    t1 = time.perf_counter()
    cl_genomes = [[[frags[i]] for i in c] for c in clusters]
    kept = []
    for g in cl_genomes:
        seqs = [s for gg in g for s in gg]
        cands = candidate_probes.candidate_strings_from_sequences(seqs, L, STRIDE)
        params = orc.minhash_draw_params(orc.minhash_num_tables(0.6), 3)
        kept.append(orc.ndf_minhash(cands, 0.6, params))
    t2 = time.perf_counter()
    exp = orc.set_cover_filter(kept, cl_genomes, 5, L, coverage=1.0,
                               cover_extension=EXT, lazy=True)
    want = sorted(set(kept[i][j] for i, ids in enumerate(exp) for j in ids))
    rec = dict(genomes=len(genomes), bases=sum(len(s) for g in genomes for s in g),
               fragments=len(frags), clusters=len(clusters),
               n_candidates=sum(len(k) for k in kept), n_probes=len(want),
               probes_sha256=hashlib.sha256("\n".join(want).encode()).hexdigest())
    sys.stderr.write("S5 x%g: %s (cluster %.0f s, ndf %.0f s, total %.0f s)\n"
                     % (scale, rec, t1 - t0, t2 - t1, time.perf_counter() - t0))
    return dict(flags="design_large defaults (-m 5 -e 50, --cluster-and-design-separately "
                      "0.15 simple, --cluster-from-fragments 50000, "
                      "--filter-with-lsh-minhash 0.6); random.seed(21), "
                      "np.random.seed(22); PYTHONHASHSEED=0; sha256 over the sorted "
                      "distinct probe strings joined by newlines",
                seed="synthetic.dataset default", design=rec)


def main():
    orc.build()
    orc.set_threads(int(os.environ.get("ORC_THREADS", str(orc.hw_threads()))))
    try:
        with open(OUT) as f:
            out = json.load(f)
    except (OSError, ValueError):
        out = {}
    for spec in sys.argv[1:]:
        name, _, sc = spec.partition(":")
        scale = float(sc) if sc else 1.0
        key = name if scale == 1.0 else "%s:%g" % (name, scale)
        if name == "S3":
            out[key] = config3(scale)
        elif name == "S5":
            out[key] = config5(scale)
        else:
            out[key] = dict(flags="-pl 100 -ps 50 -m 2 -e 50 -c 1.0, DuplicateFilter "
                              "then SetCoverFilter per group; pick ids index the "
                                  "group's de-duplicated candidates in first-occurrence "
                                  "order; sha256 over the sorted ids as little-endian int64",
                            seed="synthetic.dataset default", groups=set_cover_groups(name, scale))
        with open(OUT, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
