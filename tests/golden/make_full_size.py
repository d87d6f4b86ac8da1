#!/usr/bin/env python3
"""Digests of the selected probes at the FULL sizes of BASELINE configs[2..4].

Authoring container only (minutes to an hour of CPU).  The GPU box cannot run
the CPU oracle at these sizes inside a test, so the expected answers are
computed here -- with the oracle that tests/test_oracle_golden.py pins to the
reference (threaded per-sequence scans + the lazy evaluation of the greedy,
which that file proves equal to the line-by-line restatement) -- and committed
as data: per group (n_candidates, n_picks, sha256 of the sorted pick ids).
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

OUT = os.path.join(HERE, "full_size_picks.json")
L, STRIDE, MISMATCHES, EXT = 100, 50, 2, 50


def digest(ids):
    a = np.sort(np.asarray(ids, dtype=np.int64))
    return hashlib.sha256(a.astype("<i8").tobytes()).hexdigest()


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
        picks = orc.lazy_greedy(rows[0], rows[1], rows[2], rows[3], len(cands),
                                [sum(len(s) for s in g) for g in genomes])
        recs.append(dict(group=gi, genomes=len(genomes),
                         bases=sum(len(s) for s in seqs),
                         n_candidates=len(cands), n_rows=int(rows[0].size),
                         n_picks=len(picks), picks_sha256=digest(picks)))
        sys.stderr.write("%s x%g group %d: %s (%.0f s)\n"
                         % (name, scale, gi, recs[-1], time.perf_counter() - t0))
        sys.stderr.flush()
        del rows
    return recs


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
        out[key] = dict(flags="-pl 100 -ps 50 -m 2 -e 50 -c 1.0, DuplicateFilter "
                              "then SetCoverFilter per group; pick ids index the "
                              "group's de-duplicated candidates in first-occurrence "
                              "order; sha256 over the sorted ids as little-endian int64",
                        seed="synthetic.dataset default", groups=set_cover_groups(name, scale))
        with open(OUT, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
