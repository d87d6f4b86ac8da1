#!/usr/bin/env python3
"""GPU SetCoverFilter wall-clock (M2: candidate strings + genomes in -> selected probes out, packing and H2D included)
on the REAL Ebola inputs the LIVE reference was run on (tests/golden/real_runs.json, made by
tests/golden/make_real_golden.py in the authoring container), and the equality of the selections.
    python tools/gpu_vs_reference_real.py > profiles/r04_gpu_vs_reference_real.json"""
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
from catch_amd.filter import candidate_probes  # noqa: E402
from catch_amd.filter.set_cover_filter import SetCoverFilter  # noqa: E402
from catch_amd.utils import seq_io  # noqa: E402

here = os.path.join(REPO, "tests", "golden")
d = json.load(open(os.path.join(here, "real_runs.json")))
genomes_all = seq_io.read_genomes_from_fasta(os.path.join(here, d["fasta"]))
out = []
for r in d["runs"]:
    gen = genomes_all[:r["records"]]
    pl = r["probe_length"]
    cands = list(dict.fromkeys(candidate_probes.candidate_strings_from_sequences(
        [s for g in gen for s in g.seqs], pl, pl // 2)))
    best = None
    for rep in range(3):
        if r["np_random_seed"] is not None:
            np.random.seed(r["np_random_seed"])
        f = SetCoverFilter(mismatches=r["mismatches"], lcf_thres=r["lcf_thres"], coverage=r["coverage"],
                           cover_extension=r["cover_extension"])
        t0 = time.perf_counter()
        ids = f._filter_strs([cands], [gen], assume_unique=True)[0]
        wall = time.perf_counter() - t0
        best = wall if best is None else min(best, wall)
    sel = sorted(cands[i] for i in ids)
    dig = hashlib.sha256(",".join(sel).encode()).hexdigest()
    out.append(dict(records=r["records"], G=r["G"], P=r["P"], probe_length=pl, mismatches=r["mismatches"],
                    lcf_thres=r["lcf_thres"], coverage=r["coverage"], cover_extension=r["cover_extension"],
                    reference_wall_s=r["reference_wall_s"], gpu_wall_s=round(best, 5),
                    speedup=round(r["reference_wall_s"] / best, 1), probes_out=len(sel),
                    identical=dig == r["picks_sha256"]))
json.dump(dict(input=d["source"], runs=out), sys.stdout, indent=1)
