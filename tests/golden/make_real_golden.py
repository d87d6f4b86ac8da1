#!/usr/bin/env python3
"""Real-sequence parity fixture (authoring container only).

The only real FASTA the reference holds is its own test datum
catch/utils/tests/data/zaire_ebolavirus.fasta.gz (1,525 Ebola genomes).  This
script (i) copies the first 100 RECORDS of it -- data, not code -- to
tests/golden/ebola_zaire_100.fasta.gz, and (ii) runs the LIVE reference
(imported read-only from /root/reference) on the first 30 / 100 records:
candidate probes (stride = half the probe length) -> DuplicateFilter ->
SetCoverFilter, and records the digest of every selection in
tests/golden/real_runs.json.  Real viral composition has what the synthetic
genomes lack: low-complexity runs, repeats inside a genome, real indels and
N runs.  `-l 60` cases take the reference's random-anchor map (np.random is
seeded and consumed as the reference consumes it) and the general
(truncated-alignment) cover function.

    PYTHONHASHSEED=0 python tests/golden/make_real_golden.py [quick]
"""
import gzip
import hashlib
import json
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
from catch.filter import candidate_probes  # noqa: E402
from catch.filter import duplicate_filter  # noqa: E402
from catch.filter import set_cover_filter as scf  # noqa: E402
from catch.utils import seq_io  # noqa: E402

SRC = "/root/reference/catch/utils/tests/data/zaire_ebolavirus.fasta.gz"
OUT_FASTA = os.path.join(HERE, "ebola_zaire_100.fasta.gz")
NREC = 100


def subset():
    recs, cur = [], None
    with gzip.open(SRC, "rt") as f:
        for line in f:
            if line.startswith(">"):
                if len(recs) == NREC:
                    break
                cur = [line.rstrip("\n"), []]
                recs.append(cur)
            elif cur is not None:
                cur[1].append(line.strip())
    with gzip.GzipFile(OUT_FASTA, "wb", mtime=0) as g:
        for h, parts in recs:
            g.write((h + "\n" + "".join(parts) + "\n").encode())
    return len(recs)


def run(n, pl, mm, ext, lcf, cov, seed):
    genomes = seq_io.read_genomes_from_fasta(OUT_FASTA)[:n]
    cands = []
    for g in genomes:
        cands += candidate_probes.make_candidate_probes_from_sequences(
            g.seqs, probe_length=pl, probe_stride=pl // 2)
    cands = duplicate_filter.DuplicateFilter().filter(cands)
    if seed is not None:
        np.random.seed(seed)
    f = scf.SetCoverFilter(mismatches=mm, lcf_thres=lcf, coverage=cov, cover_extension=ext)
    t0 = time.perf_counter()
    out = f.filter([cands], [genomes], input_is_grouped=True)
    wall = time.perf_counter() - t0
    sel = sorted(p.seq_str for p in out[0])
    return dict(records=n, probe_length=pl, mismatches=mm, cover_extension=ext, lcf_thres=lcf, coverage=cov,
                np_random_seed=seed, G=sum(g.size() for g in genomes), P=len(cands), probes_out=len(sel),
                picks_sha256=hashlib.sha256(",".join(sel).encode()).hexdigest(),
                reference_wall_s=round(wall, 2))


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    n = subset()
    assert n == NREC
    specs = [(30, 100, 2, 50, 100, 1.0, None),
             (30, 75, 2, 50, 75, 1.0, None),
             (30, 75, 2, 50, 60, 1.0, 11),       # -l 60: random anchors, truncated alignments
             (30, 100, 3, 0, 100, 0.9, None)]
    if not quick:
        specs += [(100, 100, 2, 50, 100, 1.0, None),
                  (100, 75, 2, 50, 60, 1.0, 12)]
    runs = []
    for s in specs:
        r = run(*s)
        runs.append(r)
        sys.stderr.write(json.dumps(r) + "\n")
        sys.stderr.flush()
        with open(os.path.join(HERE, "real_runs.json"), "w") as f:
            json.dump(dict(fasta=os.path.basename(OUT_FASTA), source="catch/utils/tests/data/zaire_ebolavirus.fasta.gz "
                           "(first %d records)" % NREC, python=sys.version.split()[0],
                           hashseed=os.environ.get("PYTHONHASHSEED"), runs=runs), f, indent=1)


if __name__ == "__main__":
    main()
