#!/usr/bin/env python3
"""Wall-clock of the LIVE reference's SetCoverFilter on the synthetic inputs.

Authoring container only (imports /root/reference read-only; nothing from it
is copied).  For every named input the reference builds its own candidate
probes (catch.filter.candidate_probes, stride 50) + DuplicateFilter and runs
`SetCoverFilter(mismatches=2, lcf_thres=100, cover_extension=50).filter(...,
input_is_grouped=True)` with the default process count (8 here); the phases
are cut at the reference's own INFO records
(catch/filter/set_cover_filter.py:396,415,822-833,913,918).  Also records a
digest of the selected probes, which the parity tests compare with.

    PYTHONHASHSEED=0 python tools/time_reference.py S1:1 S2:1 S3:0.004 S4:0.004 \
        > profiles/r02_reference_timings.json
"""
import hashlib
import importlib.util  # noqa: F401
import json
import logging
import multiprocessing
import os
import sys
import time

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, REPO)

from catch import genome  # noqa: E402
from catch.filter import candidate_probes  # noqa: E402
from catch.filter import duplicate_filter  # noqa: E402
from catch.filter import set_cover_filter as scf  # noqa: E402

from catch_amd.utils import synthetic  # noqa: E402


class Marks(logging.Handler):
    def __init__(self):
        super().__init__(level=logging.INFO)
        self.marks = []

    def emit(self, record):
        self.marks.append((time.perf_counter(), record.getMessage()))


def run(name, scale):
    groups = synthetic.dataset(name, scale=scale)
    genomes = [[genome.Genome.from_chrs(
        dict(("c%d" % i, s) for i, s in enumerate(g))) if len(g) > 1
        else genome.Genome.from_one_seq(g[0]) for g in grp] for grp in groups]
    t0 = time.perf_counter()
    cands = []
    for grp in genomes:
        c = []
        for g in grp:
            c += candidate_probes.make_candidate_probes_from_sequences(
                g.seqs, probe_length=100, probe_stride=50)
        cands.append(duplicate_filter.DuplicateFilter().filter(c))
    t_cand = time.perf_counter() - t0
    h = Marks()
    lg = logging.getLogger("catch.filter.set_cover_filter")
    lg.setLevel(logging.INFO)
    lg.addHandler(h)
    f = scf.SetCoverFilter(mismatches=2, lcf_thres=100, coverage=1.0,
                           cover_extension=50)
    t0 = time.perf_counter()
    out = f.filter(cands, genomes, input_is_grouped=True)
    wall = time.perf_counter() - t0
    lg.removeHandler(h)
    # phases: "Building set cover sets input" .. "ranks input" = _make_sets;
    # "Solving set cover instances" .. end = pool solve
    make_sets = 0.0
    t_sets = None
    t_solve = None
    for t, m in h.marks:
        if m.startswith("Building set cover sets input"):
            t_sets = t
        elif m.startswith("Building set cover ranks input") and t_sets is not None:
            make_sets += t - t_sets
            t_sets = None
        elif m.startswith("Solving set cover instances"):
            t_solve = t
    solve = (t0 + wall - t_solve) if t_solve is not None else None
    P = [len(c) for c in cands]
    G = [sum(len(s) for g in grp for s in g) for grp in groups]
    sel = [sorted(p.seq_str for p in grp) for grp in out]
    dig = hashlib.sha256("\n".join(",".join(g) for g in sel).encode()).hexdigest()
    return dict(input=name, scale=scale, genomes=sum(len(g) for g in groups),
                groups=len(groups), G=sum(G), P=sum(P),
                units_sum_PgGg=sum(p * g for p, g in zip(P, G)),
                candidates_s=round(t_cand, 3), make_sets_s=round(make_sets, 3),
                solve_s=None if solve is None else round(solve, 3),
                setcoverfilter_wall_s=round(wall, 3),
                probes_out=sum(len(g) for g in sel), picks_sha256=dig,
                processes=min(multiprocessing.cpu_count(), 8))


def main():
    res = []
    for spec in sys.argv[1:]:
        name, scale = spec.split(":")
        r = run(name, float(scale))
        res.append(r)
        sys.stderr.write(json.dumps(r) + "\n")
        sys.stderr.flush()
    json.dump(dict(host_cpus=multiprocessing.cpu_count(),
                   python=sys.version.split()[0],
                   flags="-pl 100 -ps 50 -m 2 -e 50 -c 1.0 (DuplicateFilter, "
                         "SetCoverFilter); PYTHONHASHSEED=%s"
                         % os.environ.get("PYTHONHASHSEED"),
                   runs=res), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
