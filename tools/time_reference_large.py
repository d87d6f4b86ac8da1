#!/usr/bin/env python3
"""Wall-clock of the LIVE reference's design_large chain on S5 x small (configs[4]).

Authoring container only (imports /root/reference read-only; nothing from it is
copied).  The chain bin/design_large.py runs with its defaults (bin/design.py:
502, 583, 753, 794, 846): 50-kb fragments -> MinHash clustering at 0.15 ->
per cluster candidate probes, MinHash near-duplicate filter 0.6, SetCoverFilter
-m 5 -e 50 -> merged probes.  Seeds as in tests/golden/make_full_size.py
(random.seed(21), np.random.seed(22)); must run under PYTHONHASHSEED=0 (the
MinHash filter hashes k-mers with the interpreter's str hash).

    PYTHONHASHSEED=0 python tools/time_reference_large.py S5m:0.01 S5m:0.05 > profiles/r03_reference_timings_S5.json
"""
import hashlib
import json
import logging
import multiprocessing
import os
import random
import sys
import time

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, REPO)

import numpy as np  # noqa: E402
from catch import genome  # noqa: E402
from catch.filter import near_duplicate_filter as ndf  # noqa: E402
from catch.filter import probe_designer  # noqa: E402
from catch.filter import set_cover_filter as scf  # noqa: E402

from catch_amd.utils import synthetic  # noqa: E402


def run(spec):
    name, _, sc = spec.partition(":")
    scale = float(sc) if sc else 1.0
    genomes = synthetic.dataset(name, scale=scale)[0]
    grouped = [[genome.Genome.from_one_seq(g[0]) for g in genomes]]
    random.seed(21)
    np.random.seed(22)
    f_ndf = ndf.NearDuplicateFilterWithMinHash(0.6)
    f_scf = scf.SetCoverFilter(mismatches=5, lcf_thres=100, coverage=1.0, cover_extension=50,
                               kmer_probe_map_k=20)
    pd = probe_designer.ProbeDesigner(grouped, [f_ndf, f_scf], probe_length=100, probe_stride=50,
                                      cluster_threshold=0.15, cluster_merge_after=f_scf,
                                      cluster_method="choose", cluster_fragment_length=50000)
    t0 = time.perf_counter()
    pd.design()
    wall = time.perf_counter() - t0
    probes = sorted(set(p.seq_str for p in pd.final_probes))
    return dict(input=name, scale=scale, genomes=len(genomes), bases=sum(len(s) for g in genomes for s in g),
                design_wall_s=round(wall, 2), probes_out=len(probes),
                probes_sha256=hashlib.sha256("\n".join(probes).encode()).hexdigest(),
                processes=min(multiprocessing.cpu_count(), 8))


def main():
    logging.basicConfig(level=logging.WARNING)
    res = []
    for spec in sys.argv[1:]:
        r = run(spec)
        res.append(r)
        sys.stderr.write(json.dumps(r) + "\n")
        sys.stderr.flush()
    json.dump(dict(host_cpus=multiprocessing.cpu_count(), python=sys.version.split()[0],
                   flags="design_large defaults: -m 5 -e 50 -pl 100 -ps 50, cluster 0.15 from 50-kb fragments "
                         "(choose), MinHash near-duplicate filter 0.6; random.seed(21), np.random.seed(22); "
                         "PYTHONHASHSEED=%s" % os.environ.get("PYTHONHASHSEED"),
                   runs=res), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
