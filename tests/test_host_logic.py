"""Host-side logic of the product (no GPU): anchor tables, candidate probes,
plugin dispatch, FASTA rules, and that the C-ABI library loads and exports
every symbol include/catchhip.h declares."""
import os
import random
import re

import numpy as np
import pytest

from util import load_golden, np_state_from_json

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from catch_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(REPO, "include", "catchhip.h")).read()
    declared = set(re.findall(r"\b(catchhip_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"catchhip_ctx", "catchhip_targets", "catchhip_probes",
                 "catchhip_rows"}
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), name
        assert name in _lib.PROTOTYPES, name
    assert set(_lib.PROTOTYPES) == declared
    assert L.catchhip_abi_version() == 1


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product raises; it never routes through the oracle."""
    from catch_amd import engine, _lib
    if engine.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_lib.CatchHipError):
        engine.Context(0)
    import catch_amd.filter.set_cover_filter as scf
    src = open(scf.__file__).read() + open(engine.__file__).read()
    assert "oracle" not in src


def test_anchor_table_matches_reference_rows_inputs(oracle):
    """anchor_table (product) == oracle's restatement, pigeonhole and random
    (np.random consumed identically), duplicates pooled."""
    from catch_amd import probe
    rng = random.Random(5)
    for L, m, thres, min_k in [(75, 2, 75, 20), (100, 2, 100, 20),
                               (100, 5, 100, 20), (75, 2, 60, 20),
                               (100, 0, 100, 20), (100, 4, 100, 20),
                               (30, 1, 30, 3), (12, 3, 12, 3)]:
        strs = ["".join(rng.choice("ACGT") for _ in range(L)) for _ in range(40)]
        strs += strs[:5]          # duplicates
        np.random.seed(42)
        k1, uniq, owner, ep, eo = probe.anchor_table(strs, m, thres, min_k, min_k)
        state_after = np.random.get_state()[2]
        np.random.seed(42)
        k2, entries = oracle.anchor_table(strs, m, thres, min_k, min_k)
        assert np.random.get_state()[2] == state_after     # the generator was advanced identically
        u2, own2 = oracle._unique_last(strs)
        assert k1 == k2 and uniq == u2 and list(owner) == own2
        assert sorted(zip(ep.tolist(), eo.tolist())) == entries
    # probes of different lengths (always random anchors), draw order included
    mixed = ["".join(rng.choice("ACGT") for _ in range(rng.choice([40, 40, 55, 90]))) for _ in range(60)]
    mixed += mixed[:4]
    np.random.seed(7)
    k1, uniq, owner, ep, eo, draws = probe.anchor_table(mixed, 2, 40, 20, 20, with_draws=True)
    np.random.seed(7)
    k2, entries, draws2 = oracle.anchor_table(mixed, 2, 40, 20, 20, with_draws=True)
    assert k1 == k2 and list(zip(ep.tolist(), eo.tolist())) == entries and draws == draws2



def test_pigeonhole_kmer_length_table():
    """SURVEY App. A.2 table (measured on the reference)."""
    from catch_amd import probe
    assert probe.pigeonhole_kmer_length(75, 0) == 75
    assert probe.pigeonhole_kmer_length(75, 1) == 25
    assert probe.pigeonhole_kmer_length(75, 2) == 25
    assert probe.pigeonhole_kmer_length(100, 1) == 50
    assert probe.pigeonhole_kmer_length(100, 2) == 25
    assert probe.pigeonhole_kmer_length(100, 3) == 25
    assert probe.pigeonhole_kmer_length(100, 4) == 20
    assert probe.pigeonhole_kmer_length(100, 5) < 20


def test_candidate_probes_golden():
    from catch_amd.filter import candidate_probes
    recs = load_golden("candidate_probes")
    assert recs
    for c in recs:
        got = candidate_probes.make_candidate_probes_from_sequences(
            c["seqs"], c["probe_length"], c["probe_stride"])
        assert [p.seq_str for p in got] == c["out"]
        # the string fast path of the front end gives the same list
        assert candidate_probes.candidate_strings_from_sequences(
            c["seqs"], c["probe_length"], c["probe_stride"]) == c["out"]
    import random
    rnd = random.Random(7)
    for _ in range(200):   # N runs, ends that are not a multiple of the stride, short tails
        n = rnd.randrange(30, 400)
        seq = "".join(rnd.choice("ACGTN" if rnd.random() < 0.3 else "ACGT")
                      for _ in range(n))
        if rnd.random() < 0.5:
            i = rnd.randrange(0, n)
            seq = seq[:i] + "N" * rnd.randrange(1, 6) + seq[i:]
        L, st = rnd.choice([(20, 7), (30, 15), (25, 25)])
        want = [p.seq_str for p in
                candidate_probes.make_candidate_probes_from_sequences([seq], L, st)]
        assert candidate_probes.candidate_strings_from_sequences([seq], L, st) == want
    with pytest.raises(ValueError):
        candidate_probes.make_candidate_probes_from_sequences(["ACGT"], 10, 5)
    with pytest.raises(ValueError):
        candidate_probes.candidate_strings_from_sequences(["ACGT"], 10, 5)
    with pytest.raises(TypeError):
        candidate_probes.make_candidate_probes_from_sequences("ACGT", 2, 1)


def test_base_filter_dispatch():
    from catch_amd.filter.base_filter import BaseFilter

    class One(BaseFilter):
        def _filter(self, input):
            return [x for x in input if x % 2 == 0]

    class Two(BaseFilter):
        def _filter(self, input, target_genomes):
            return [x for x in input if x in target_genomes]

    class Grouped(BaseFilter):
        requires_probe_groupings = True

        def _filter(self, input, target_genomes):
            return [list(reversed(g)) for g in input]

    assert One().filter([1, 2, 3, 4]) == [2, 4]
    assert One().filter([[1, 2], [4, 5]], input_is_grouped=True) == [[2], [4]]
    assert Two().filter([1, 2, 3], [2, 3]) == [2, 3]
    assert Grouped().filter([[1, 2], [3]], [[], []], input_is_grouped=True) == [[2, 1], [3]]
    with pytest.raises(AssertionError):
        Grouped().filter([1, 2], [[]])
    with pytest.raises(Exception):
        BaseFilter().filter([1])


def test_duplicate_filter_and_probe_type():
    from catch_amd import probe
    from catch_amd.filter.duplicate_filter import DuplicateFilter
    ps = [probe.Probe.from_str(s) for s in ["ACGT", "TTTT", "ACGT", "GGGG", "TTTT"]]
    out = DuplicateFilter().filter(ps)
    assert [p.seq_str for p in out] == ["ACGT", "TTTT", "GGGG"]
    assert out[0] is ps[0]
    a, b = probe.Probe.from_str("ACGTN"), probe.Probe.from_str("ACCTN")
    assert a.mismatches(b) == 1 and len(a) == 5 and a[1] == "C"
    assert "".join(a.seq) == "ACGTN"
    assert a.reverse_complement().seq_str == "NACGT"
    with pytest.raises(ValueError):
        a.mismatches(probe.Probe.from_str("AC"))


def test_seq_io_rules(tmp_path):
    from catch_amd.utils import seq_io
    fn = tmp_path / "x.fasta"
    fn.write_text(">a desc\nacgtRYn-\nAC-GT\n\n>b\nNNNN\nacgu\n")
    m = seq_io.read_fasta(str(fn))
    assert list(m.items()) == [("a desc", "ACGTNNNACGT"), ("b", "NNNNACGU")]
    assert list(seq_io.iterate_fasta(str(fn))) == ["acgtNNn-AC-GT", "NNNNacgu"]
    gs = seq_io.read_genomes_from_fasta(str(fn))
    assert [g.seqs for g in gs] == [["ACGTNNNACGT"], ["NNNNACGU"]]
    assert gs[0].size() == 11


def test_ndf_ordering_and_positions():
    from catch_amd import probe
    from catch_amd.filter import near_duplicate_filter as ndf
    g = load_golden("ndf_hamming")
    for c in g["synthetic"]:
        f = ndf.NearDuplicateFilterWithHammingDistance(c["dist_thres"], c["dim"])
        assert f.num_tables() == len(c["positions"])
        random.seed(c["seed"])
        assert f._draw_positions() == c["positions"]
    f = ndf.NearDuplicateFilterWithHammingDistance(2, 4)
    ps = [probe.Probe.from_str(s) for s in ["AAAA", "CCCC", "CCCC", "GGGG", "GGGG", "TTTT"]]
    assert [p.seq_str for p in f._order_by_multiplicity(ps)] == ["CCCC", "GGGG", "AAAA", "TTTT"]


def test_set_cover_filter_constructor_surface():
    from catch_amd.filter import set_cover_filter as scf
    f = scf.SetCoverFilter(2, 100, coverage=0.5, cover_extension=10)
    assert f.requires_probe_groupings is True
    assert (f.mismatches_tolerant, f.lcf_thres_tolerant) == (2, 100)
    assert f.filter([[]], [[]], input_is_grouped=True) == [[]]
    with pytest.raises(NotImplementedError):
        scf.SetCoverFilter(2, 100, custom_cover_range_fn=("x.py", "fn"))
    scf.set_max_num_processes_for_set_cover_instances(4)
    from catch_amd.genome import Genome
    g = [Genome.from_one_seq("A" * 50), Genome.from_one_seq("C" * 200)]
    assert scf.SetCoverFilter(2, 100, coverage=100)._make_universe_p(g) == [1.0, 0.5]
    assert scf.SetCoverFilter(2, 100, coverage=0.25)._make_universe_p(g) == [0.25, 0.25]


def test_multiplicity_order_on_strings_matches_objects():
    """The string pipeline's multiplicity order (Counter + stable sort) is the
    object pipeline's (near_duplicate_filter.py:60-66)."""
    import random
    from catch_amd import probe
    from catch_amd.filter import near_duplicate_filter as ndf
    rnd = random.Random(3)
    pool = ["".join(rnd.choice("ACGT") for _ in range(12)) for _ in range(40)]
    strs = [rnd.choice(pool) for _ in range(500)]
    f = ndf.NearDuplicateFilterWithHammingDistance(1, 12)
    objs = f._order_by_multiplicity([probe.Probe.from_str(s) for s in strs])
    assert ndf._order_strs_by_multiplicity(strs) == [p.seq_str for p in objs]
    assert ndf.NearDuplicateFilterWithMinHash(0.6).num_tables() == 25
    assert ndf.NearDuplicateFilterWithHammingDistance(2, 100).num_tables() == 2


def test_anchor_table_assume_unique_matches_general():
    from catch_amd import probe
    import random
    rnd = random.Random(5)
    strs = list(dict.fromkeys("".join(rnd.choice("ACGT") for _ in range(100))
                              for _ in range(50)))
    for m, thres in ((2, 100), (5, 100), (2, 60)):
        np.random.seed(4)
        a = probe.anchor_table(strs, m, thres)
        np.random.seed(4)
        b = probe.anchor_table(strs, m, thres, assume_unique=True)
        assert a[0] == b[0] and a[1] == b[1]
        for x, y in zip(a[2:], b[2:]):
            assert np.array_equal(x, y)


def test_connected_components_and_linkage_golden():
    """Host side of the clustering pre-step (catch_amd/utils/cluster.py): the
    depth-first search with the early-stop heuristic and the hierarchical
    cut, on distance functions recorded from the reference's tests and on
    random graphs where the visiting order decides the outcome."""
    import sys
    from catch_amd.utils import cluster
    g = load_golden("cluster")

    def dist_of(rows):
        return lambda i, j: rows[min(i, j)][max(i, j) - min(i, j) - 1]
    t = g["from_reference_tests"]
    for c in t["cc"]:
        kw = ({"early_stop_threshold": c["early_stop_threshold"]}
              if "early_stop_threshold" in c else {})
        assert cluster.find_connected_components(
            c["n"], dist_of(c["dist"]), c["threshold"], **kw) == c["out"]
    for c in t["hier"]:
        assert cluster.cluster_hierarchically_from_dist_matrix(
            np.asarray(c["dist_matrix"], dtype=np.float32), c["threshold"]) == c["out"]
    same_python = g["python"].split(".")[:2] == sys.version.split()[0].split(".")[:2]
    vals = (0.05, 0.3, 0.9)
    for c in g["cc_stress"]:
        n, cls = c["n"], c["classes"]
        rows, at = [], 0
        for i in range(n):
            rows.append([vals[int(x)] for x in cls[at:at + n - i - 1]])
            at += n - i - 1
        got = cluster.find_connected_components(n, dist_of(rows), c["threshold"],
                                                c["early_stop_threshold"])
        if same_python:
            assert got == c["out"]
        assert sorted(x for comp in got for x in comp) == list(range(n))
    assert cluster.find_connected_components(0, None, 1) == []
    m = cluster.create_condensed_dist_matrix(3, lambda i, j: [[0, 1, 100], [1, 0, 2], [100, 2, 0]][i][j])
    assert m.dtype == np.float32 and m.tolist() == [1.0, 100.0, 2.0]
    assert abs(cluster._jaccard_dist_from_mash_dist(0.1, 12) - (1.0 - 1.0 / (2.0 * np.exp(12 * 0.1) - 1))) == 0


def test_genome_fragments():
    from collections import OrderedDict
    from catch_amd.genome import Genome
    g = Genome.from_one_seq("ABCDEFGHIJ")
    assert g.break_into_fragments(4).seqs == ["ABCD", "EFGH", "IJ"]
    assert g.break_into_fragments(4, include_full_end=True).seqs == ["ABCD", "EFGH", "GHIJ"]
    assert g.break_into_fragments(20, include_full_end=True).seqs == ["ABCDEFGHIJ"]
    h = Genome.from_chrs(OrderedDict([("x", "AAAAAB"), ("y", "CC")]))
    f = h.break_into_fragments(3, include_full_end=True)
    assert list(f.chrs.items()) == [("x-0", "AAA"), ("x-1", "AAB"), ("y-0", "CC")]


def test_anchor_entries_without_strings_match_anchor_table():
    """probe.anchor_entries_equal_length (the device front end never sees the
    candidate strings) == anchor_table on distinct strings of that length:
    same rule, same entries, same np.random stream."""
    from catch_amd import probe
    rng = random.Random(9)
    for L, m, thres, min_k in [(100, 2, 100, 20), (100, 5, 100, 20), (75, 2, 60, 20),
                               (100, 0, 100, 20), (60, 3, 60, 10), (30, 1, 30, 20)]:
        strs = list(dict.fromkeys("".join(rng.choice("ACGT") for _ in range(L)) for _ in range(70)))
        np.random.seed(12)
        k1, _u, _o, ep, eo = probe.anchor_table(strs, m, thres, min_k, min_k, assume_unique=True)
        state1 = np.random.get_state()[2]
        np.random.seed(12)
        k2, ep2, eo2 = probe.anchor_entries_equal_length(len(strs), L, m, thres, min_k, min_k)
        assert k1 == k2 and np.random.get_state()[2] == state1
        if ep2 is None:      # pigeonhole table {0, k, 2k, ..}: generated on the device
            assert eo.tolist() == list(range(0, L, k1)) * len(strs)
        else:
            assert np.array_equal(ep, ep2) and np.array_equal(eo, eo2)
    assert probe.anchor_entries_equal_length(0, 100, 5, 100)[1].size == 0


def test_probe_designer_object_pipeline_without_gpu():
    """Filter lists other than [duplicate / near-duplicate filter, set cover]
    run as objects through BaseFilter.filter: contrived inputs in the style of
    the reference's designer tests (probe length 100 / stride 50; short
    sequences with allow_small_seqs)."""
    from catch_amd import genome
    from catch_amd.filter import duplicate_filter, probe_designer
    seqs = [[genome.Genome.from_one_seq("A" * 100 + "C" * 100 + "A" * 100)]]
    pd = probe_designer.ProbeDesigner(seqs, [duplicate_filter.DuplicateFilter()],
                                      probe_length=100, probe_stride=50)
    pd.design()
    assert [p.seq_str for p in pd.candidate_probes] == [
        "A" * 100, "A" * 50 + "C" * 50, "C" * 100, "C" * 50 + "A" * 50, "A" * 100]
    assert sorted(p.seq_str for p in pd.final_probes) == sorted(
        ["A" * 100, "A" * 50 + "C" * 50, "C" * 100, "C" * 50 + "A" * 50])
    two = [[genome.Genome.from_one_seq("A" * 200), genome.Genome.from_one_seq("G" * 150)],
           [genome.Genome.from_one_seq("C" * 300)]]
    pd = probe_designer.ProbeDesigner(two, [duplicate_filter.DuplicateFilter()],
                                      probe_length=100, probe_stride=50)
    pd.design()
    assert len(pd.candidate_probes) == 10
    assert sorted(p.seq_str for p in pd.final_probes) == ["A" * 100, "C" * 100, "G" * 100]
    small = [[genome.Genome.from_one_seq("ACGTTGCAACGGTA"), genome.Genome.from_one_seq("TTGAC")]]
    pd = probe_designer.ProbeDesigner(small, [duplicate_filter.DuplicateFilter()], probe_length=6,
                                      probe_stride=3, allow_small_seqs=5)
    pd.design()
    assert [p.seq_str for p in pd.final_probes] == ["ACGTTG", "TTGCAA", "CAACGG", "ACGGTA", "TTGAC"]
    with pytest.raises(ValueError):
        probe_designer.ProbeDesigner(small, [duplicate_filter.DuplicateFilter()], probe_length=6,
                                     probe_stride=3).design()


def test_prefetch_order_errors_and_discard():
    """engine.Prefetch (the helper thread that packs group i + 1 while group i
    computes): results arrive in order, at most `depth` ahead; an exception in
    build() surfaces at the consumer; results built but never consumed are
    handed to `discard`."""
    import threading
    import time
    from catch_amd import engine
    built, dropped = [], []
    lock = threading.Lock()

    def build(i):
        with lock:
            built.append(i)
        if i == 7:
            raise ValueError("boom")
        return i * i

    pre = engine.Prefetch(range(5), build, depth=2, discard=dropped.append)
    got = list(pre)
    pre.close()
    assert got == [(i, i * i) for i in range(5)] and dropped == []

    # bounded run-ahead: after the consumer took item 0, the helper may have
    # built at most items 1..3 (two queued + one waiting to be queued)
    del built[:]
    pre = engine.Prefetch(range(10), build, depth=2, discard=dropped.append)
    it = iter(pre)
    assert next(it) == (0, 0)
    time.sleep(0.3)
    with lock:
        assert max(built) <= 3
    with pytest.raises(ValueError):
        for _ in it:
            pass
    pre.close()
    assert 7 in built and 8 not in built

    # stopped early: what was built and not consumed is discarded exactly once
    del built[:], dropped[:]
    pre = engine.Prefetch(range(6), lambda i: i + 100, depth=2, discard=dropped.append)
    it = iter(pre)
    assert next(it) == (0, 100)
    time.sleep(0.2)
    pre.close()
    assert sorted(dropped) == sorted(set(dropped)) and all(100 < d < 106 for d in dropped) and len(dropped) >= 2


def test_components_search_shortcut_equals_the_set_order():
    """The connected-components search with device-made neighbour lists
    (catch_amd/utils/cluster.py: while `remaining - queued` is known to iterate
    in ascending order the difference is never built) == the search that builds
    every difference and follows the interpreter's iteration order, on graphs
    where the early-stop heuristic makes the visiting order matter; and the
    table-size model behind it, against this interpreter's sets."""
    import sys
    from catch_amd.utils import cluster
    assert cluster._fast_order_available()
    # the model of CPython's table growth: sys.getsizeof gives the real slot count (16 bytes each
    # beyond the 8 slots inside the object)
    base = sys.getsizeof(set())
    for m in (0, 3, 4, 5, 18, 19, 20, 75, 76, 77, 307, 1228, 1229, 4915, 19660, 19661, 50000, 78643, 78644, 200000):
        s = set()
        for i in range(m):
            s.add(3 * i + 1)
        slots = 8 if sys.getsizeof(s) == base else (sys.getsizeof(s) - base) // 16
        assert cluster._table_size_after_inserts(m) == slots, m
    rng = np.random.Generator(np.random.PCG64(77))
    # whenever the predicate says so, the difference really iterates in ascending order
    for n in (12, 100, 1000, 20000):
        remaining = set(range(n))
        for _ in range(60):
            m = len(remaining)
            if m < 2:
                break
            q = int(rng.integers(1, max(2, m // int(rng.choice([2, 3, 5, 9, 40])))))
            members = rng.choice(np.fromiter(remaining, dtype=np.int64), size=q, replace=False).tolist()
            queued = set()
            for k in members:
                queued.add(k)
            if cluster._diff_iterates_ascending(n, m, len(queued)):
                d = list(remaining - queued)
                assert d == sorted(d), (n, m, q)
            remaining -= set(members[:max(1, q // 2)])
    # whole searches: random geometric-ish graphs with near (absorbing) and far (explored) edges
    used_fast = 0
    before = dict(cluster._path_counts)
    for trial in range(40):
        n = int(rng.integers(30, 400)) if trial < 25 else int(rng.integers(1500, 5000))
        x = rng.random(n) * rng.choice([3.0, 10.0, 40.0])
        dist = np.abs(x[:, None] - x[None, :]) * rng.uniform(0.5, 1.5, size=(n, n))
        dist = np.minimum(dist, dist.T)
        thr, early = 0.3, 0.08

        def row(j, cand):
            return dist[j, cand]
        calls = []

        def neighbors(j):
            calls.append(j)
            idx = np.nonzero(dist[j] <= thr)[0]
            return idx.astype(np.int64), dist[j, idx]
        def neighbors_many(js):
            assert 1 <= len(js) <= 8 and len(set(js)) == len(js)
            return [neighbors(j) for j in js]
        slow = cluster._components(n, row, thr, early)
        fast = cluster._components(n, row, thr, early, neighbors)
        assert fast == slow, trial
        used_fast += len(calls)
        # the lists of several vertices per call (the explored vertex + the top of the stack)
        asked = cluster._path_counts["list calls"]
        assert cluster._components(n, row, thr, early, neighbors, neighbors_many, batch=8) == slow, trial
        asked = cluster._path_counts["list calls"] - asked
        assert asked <= len(calls)
    assert used_fast > 1000
    took = {k: cluster._path_counts[k] - before[k] for k in before}
    # all three ways of ordering a vertex's neighbours were exercised
    assert took["ascending"] > 1000 and took["copy rank"] > 200 and took["real difference"] > 1000, took


def _planted_graph(seed, n, ncl, p, contiguous):
    """CSR graph of planted clusters (random labels, or -- as the fragments of a species lie -- contiguous blocks with
    a few strays), each pair of a cluster joined with probability p, 9-100 common values per edge."""
    rng = np.random.RandomState(seed)
    if contiguous:
        label = np.sort(rng.randint(0, ncl, size=n))
        stray = rng.random_sample(n) < 0.02
        label[stray] = rng.randint(0, ncl, size=int(stray.sum()))
    else:
        label = rng.randint(0, ncl, size=n)
    rows, cols, com = [], [], []
    for c in range(ncl):
        mem = np.nonzero(label == c)[0]
        if mem.size < 2:
            continue
        a, b = np.meshgrid(mem, mem, indexing="ij")
        keep = (a < b) & (rng.random_sample(a.shape) < p)
        aa, bb = a[keep], b[keep]
        cc = rng.randint(9, 101, size=aa.size)
        rows += [aa, bb]; cols += [bb, aa]; com += [cc, cc]
    if rows:
        rows, cols, com = np.concatenate(rows), np.concatenate(cols), np.concatenate(com)
        o = np.lexsort((cols, rows))
        rows, cols, com = rows[o], cols[o], com[o]
    else:
        rows = cols = com = np.zeros(0, dtype=np.int64)
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=ptr[1:])
    return ptr, cols.astype(np.uint32), com.astype(np.uint32)


def test_emulated_int_set_equals_the_interpreters():
    """PyIntSet (catch_amd/csrc/components.hip) == this interpreter's set of small ints, slot for slot:
    set(range(n)), `-=` (dummies, the rebuild beyond mask / 4 of them), copy(), and `a - b` in both of CPython's
    forms, by iteration order.  The product runs a shorter version of this once per process."""
    import ctypes
    from catch_amd import _lib
    from catch_amd.utils import cluster
    assert cluster._native_sets_available()
    L = _lib.lib()
    p32, cnt = _lib.c_u32p(), ctypes.c_int64(0)

    def listed(h, which, keys=()):
        k = np.ascontiguousarray(np.fromiter(keys, dtype=np.uint32, count=len(keys)))
        _lib.check(L.catchhip_pyintset_list(h, which, k.ctypes.data_as(_lib.c_u32p), int(k.size),
                                            ctypes.byref(p32), ctypes.byref(cnt)))
        return np.ctypeslib.as_array(p32, shape=(cnt.value,)).tolist() if cnt.value else []
    rs = np.random.RandomState(5)
    cases = 0
    for n in (1, 5, 9, 40, 300, 2500, 70000, 130000):
        h = ctypes.c_void_p()
        _lib.check(L.catchhip_pyintset_create(n, ctypes.byref(h)))
        remaining = set(range(n))
        for it in range(25 if n < 100000 else 8):
            m = len(remaining)
            if m == 0:
                break
            members = np.fromiter(remaining, dtype=np.int64, count=m)
            assert listed(h, 0) == members.tolist(), (n, it)
            assert listed(h, 1) == list(remaining.copy()), (n, it)
            for frac in (0.01, 0.2, 0.3, 0.6, 0.95):
                queued = set()
                for k in members[rs.permutation(m)[:max(1, min(m, int(m * frac)))]].tolist():
                    queued.add(k)
                assert listed(h, 2, queued) == list(remaining - queued), (n, it, frac)
                cases += 1
            if rs.random_sample() < 0.5:      # a contiguous stretch (a species' fragments), or scattered members
                a = int(members[rs.randint(m)])
                cc = set(members[(members >= a) & (members < a + max(1, m // int(rs.choice((3, 7, 20)))))].tolist())
            else:
                cc = set(members[rs.permutation(m)[:max(1, m // int(rs.choice((2, 5, 11))))]].tolist())
            remaining -= cc
            k = np.fromiter(cc, dtype=np.uint32, count=len(cc))
            _lib.check(L.catchhip_pyintset_isub(h, k.ctypes.data_as(_lib.c_u32p), int(k.size)))
        L.catchhip_pyintset_destroy(h)
    assert cases > 500


def test_native_components_search_equals_the_interpreter_on_random_graphs(monkeypatch):
    """catchhip_dfs_run_all (catch_amd/csrc/components.hip: the whole search natively, `remaining` emulated slot for
    slot) == the step-wise native path (catchhip_dfs_run: real sets in the interpreter wherever a layout matters)
    == _components (the same search in the interpreter) on planted-cluster graphs large enough for all three
    cases -- ascending differences, differences that are copies of `remaining`, differences built insert by insert
    -- with early-stop absorption: components and case counts; two graphs of 60,000 vertices (VERDICT round 5
    item 2: at least 50 k vertices with all three order cases), clusters scattered over the ids (every order
    simulated) and lying in blocks (orders read off the home slots).  No GPU: host code only."""
    from catch_amd.utils import cluster
    assert cluster._fast_order_available() and cluster._native_sets_available()
    for seed, n, ncl, p, contiguous in ((1, 2500, 40, 0.5, False), (2, 9000, 300, 0.5, False), (3, 700, 3, 0.5, False),
                                        (4, 1, 1, 0.5, False), (5, 4000, 4000, 0.5, False), (6, 9000, 120, 0.4, True),
                                        (11, 60000, 1500, 0.3, False), (12, 60000, 40, 0.02, False),
                                        (13, 60000, 700, 0.3, True)):
        ptr, gidx, gcom = _planted_graph(seed, n, ncl, p, contiguous)
        N = 100.0
        lut = 1.0 - np.arange(101, dtype=np.float64) / N
        threshold, early = lut[9], lut[60]
        far = np.full(n, 2.0)

        def neighbors(j):
            return gidx[ptr[j]:ptr[j + 1]], lut[gcom[ptr[j]:ptr[j + 1]]]

        def row(j, cand):
            nb = gidx[ptr[j]:ptr[j + 1]]
            far[nb] = lut[gcom[ptr[j]:ptr[j + 1]]]
            d = far[cand]
            far[nb] = 2.0
            return d
        before = dict(cluster._path_counts)
        want = cluster._components(n, row, threshold, early, neighbors, None, local_lists=True)
        took_py = {k: cluster._path_counts[k] - before[k] for k in before}
        before = dict(cluster._path_counts)
        stats0 = dict(cluster._native_stats)
        got = cluster._components_over_graph(n, ptr, gidx, gcom, 60, row, threshold, early)
        took = {k: cluster._path_counts[k] - before[k] for k in before}
        stats = {k: v - stats0.get(k, 0) for k, v in cluster._native_stats.items()}
        assert got == want, (seed, n)
        assert took == took_py, (took, took_py)
        assert stats["searches"] == 1
        monkeypatch.setenv("CATCHHIP_TEST_HOOKS", "1")
        monkeypatch.setenv("CATCHHIP_CLUSTER_STEPWISE", "1")
        before = dict(cluster._path_counts)
        step = cluster._components_over_graph(n, ptr, gidx, gcom, 60, row, threshold, early)
        took_step = {k: cluster._path_counts[k] - before[k] for k in before}
        monkeypatch.delenv("CATCHHIP_CLUSTER_STEPWISE")
        assert step == want and took_step == took_py, (seed, n)
        if n >= 2500 and ncl < n:
            assert took["copy rank"] > 0 and took["real difference"] > 0 and took["ascending"] > 0, took
        if n >= 9000 and not contiguous:
            assert stats["copies"] > 0 and stats["differences built"] > 0, stats
        if n >= 9000 and contiguous:
            assert stats["home-slot orders"] > 0, stats


def test_prefetch_pool_hands_results_over_in_order_and_discards_what_is_left():
    """engine.PrefetchPool: several builders, results in item order, bounded run-ahead, an exception of a
    builder re-raised at the consumer, built-but-unconsumed results handed to `discard` on close."""
    import threading
    import time
    from catch_amd import engine
    lock, running, peak, built = threading.Lock(), [0], [0], []

    def build(item, worker):
        with lock:
            running[0] += 1
            peak[0] = max(peak[0], running[0])
        time.sleep(0.002 * ((item * 7) % 5))
        with lock:
            running[0] -= 1
            built.append(item)
        if item == 13:
            raise KeyError("thirteen")
        return ("built", item, worker)
    gone = []
    pool = engine.PrefetchPool(range(10), build, workers=3, depth=1, discard=gone.append)
    got = [(it, res[1]) for it, res in pool]
    pool.close()
    assert got == [(i, i) for i in range(10)] and gone == [] and 1 <= peak[0] <= 3
    # run-ahead is bounded: a consumer that stops early leaves at most workers + depth results behind
    built.clear()
    pool = engine.PrefetchPool(range(100), build, workers=2, depth=1, discard=gone.append)
    it = iter(pool)
    first = [next(it)[0] for _ in range(4)]
    time.sleep(0.1)
    pool.close()
    assert first == [0, 1, 2, 3] and len(built) <= 4 + 3 and sorted(g[1] for g in gone) == sorted(set(built) - set(first))
    # a builder's exception reaches the consumer
    pool = engine.PrefetchPool(range(10, 20), build, workers=2, depth=1, discard=gone.append)
    seen = []
    with pytest.raises(KeyError):
        for it_, res in pool:
            seen.append(it_)
    pool.close()
    assert seen == [10, 11, 12]


def test_minhash_parameter_draws_in_bulk_equal_the_interpreter():
    """NearDuplicateFilterWithMinHash._draw_for_groups for many groups (a copy of
    `random`'s Mersenne Twister parsed in NumPy) == one _draw_params call per
    group, values and the state `random` is left in; few groups take the plain
    calls."""
    import random
    from catch_amd.filter.near_duplicate_filter import (NearDuplicateFilterWithMinHash,
                                                        _randint_pairs_like_random)
    f = NearDuplicateFilterWithMinHash(0.6)
    for seed, groups in ((21, 300), (3, 57), (99, 2)):
        random.seed(seed)
        want = [f._draw_params() for _ in range(groups)]
        after = random.random()
        random.seed(seed)
        got = f._draw_for_groups(groups)
        assert got == want and random.random() == after
    P = 2 ** 31 - 1
    for seed in range(5):
        random.seed(1000 + seed)
        want = [(random.randint(1, P), random.randint(0, P)) for _ in range(5000)]
        st = random.getstate()
        random.seed(1000 + seed)
        assert _randint_pairs_like_random(5000, P) == want and random.getstate() == st
    assert _randint_pairs_like_random(10, 2 ** 31 - 2) is None       # (another modulus: not what this parses)
