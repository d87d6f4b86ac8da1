#!/usr/bin/env python3
"""Near-duplicate filter -> set cover filter chains recorded from the LIVE reference.

Authoring container only (imports /root/reference read-only; nothing from it is
copied).  The reference's near-duplicate filters return `list(to_include)` -- a
SET of Probe objects -- and SetCoverFilter numbers its candidates in the order it
receives them: the selection depends on the set's iteration order, i.e. on
hash(seq_str).  Must run under PYTHONHASHSEED=0 (re-executes itself if not).
The filter is called directly (`_filter`, as `filter(..., input_is_grouped=False)`
does): over grouped input the reference forks a pool whose workers re-seed
`random` from the OS (random registers an at-fork handler), so its LSH hash
functions -- and with them the output of design.py with an LSH filter -- differ from
run to run whatever is seeded.

    python tests/golden/make_chain_golden.py  -> tests/golden/ndf_scf_chains.json
"""
import hashlib
import json
import os
import random
import sys

if os.environ.get("PYTHONHASHSEED") != "0":
    os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, PYTHONHASHSEED="0"))

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, REPO)

import numpy as np  # noqa: E402
from catch import genome  # noqa: E402
from catch.filter import candidate_probes, near_duplicate_filter as ndf, set_cover_filter as scf  # noqa: E402

from catch_amd.utils import synthetic  # noqa: E402


def sha(strs):
    return hashlib.sha256("\n".join(strs).encode()).hexdigest()


def one(name, scale, group, kind, thres, mismatches, seed):
    genomes = synthetic.dataset(name, scale=scale)[group]
    gobjs = [genome.Genome.from_chrs(dict(("c%d" % i, s) for i, s in enumerate(g))) if len(g) > 1
             else genome.Genome.from_one_seq(g[0]) for g in genomes]
    cands = []
    for g in gobjs:
        cands += candidate_probes.make_candidate_probes_from_sequences(g.seqs, probe_length=100, probe_stride=50)
    random.seed(seed)
    np.random.seed(seed + 1)
    f = (ndf.NearDuplicateFilterWithHammingDistance(thres, 100) if kind == "hamming"
         else ndf.NearDuplicateFilterWithMinHash(thres))
    kept = f._filter(cands)
    s = scf.SetCoverFilter(mismatches=mismatches, lcf_thres=100, coverage=1.0, cover_extension=50)
    out = s.filter([kept], [gobjs], input_is_grouped=True)[0]
    rec = dict(dataset=name, scale=scale, group=group, filter=kind, threshold=thres, mismatches=mismatches,
               seed=seed, genomes=len(genomes), candidates=len(cands), kept=len(kept),
               kept_in_order_sha256=sha([p.seq_str for p in kept]),
               kept_sorted_sha256=sha(sorted(p.seq_str for p in kept)),
               picks=len(out), picks_sorted_sha256=sha(sorted(p.seq_str for p in out)))
    sys.stderr.write(json.dumps(rec) + "\n")
    return rec


def main():
    runs = [one("S3", 0.004, 0, "hamming", 2, 2, 31),
            one("S3", 0.004, 0, "minhash", 0.6, 2, 32),
            one("S5m", 0.01, 0, "minhash", 0.6, 5, 33) if False else None,
            one("S4", 0.004, 7, "hamming", 3, 2, 34),
            one("S4i", 0.01, 0, "minhash", 0.5, 3, 35)]
    runs = [r for r in runs if r]
    with open(os.path.join(HERE, "ndf_scf_chains.json"), "w") as f:
        json.dump(dict(python=sys.version.split()[0], hashseed=os.environ.get("PYTHONHASHSEED"),
                       flags="-pl 100 -ps 50 -e 50 -c 1.0; the near-duplicate filter's _filter on the group's "
                             "candidates (duplicates included), then SetCoverFilter on what it returns",
                       runs=runs), f, indent=1)


if __name__ == "__main__":
    main()
