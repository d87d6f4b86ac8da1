"""Shared helpers for the tests (golden fixture loading, row canonical form)."""
import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".json.gz")
    with gzip.open(path, "rb") as f:
        return json.loads(f.read().decode())


def rows_as_tuples(set_id, univ, start, end):
    return sorted(zip((int(x) for x in set_id), (int(x) for x in univ),
                      (int(x) for x in start), (int(x) for x in end)))


def np_state_from_json(js):
    return (js[0], np.array(js[1], dtype=np.uint32), js[2], js[3], js[4])


def small_species(seed=77, n=6, length=3000, d1=0.04, d2=0.01, with_n=True):
    from catch_amd.utils import synthetic
    rng = np.random.Generator(np.random.PCG64(seed))
    return synthetic.make_species(rng, [length], n, 2, d1, d2, with_n=with_n)


def candidates(genomes, L, stride, dedup=True):
    """Candidate probe strings of one group (product host code)."""
    from catch_amd.filter import candidate_probes
    out = []
    for g in genomes:
        out += [p.seq_str for p in
                candidate_probes.make_candidate_probes_from_sequences(
                    list(g), probe_length=L, probe_stride=stride)]
    if dedup:
        out = list(dict.fromkeys(out))
    return out
