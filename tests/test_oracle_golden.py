"""Pins the CPU oracle (oracle/) to golden vectors recorded from the live
reference (tests/golden/make_golden.py).  No GPU needed."""
import os
import numpy as np
import pytest

from util import load_golden, np_state_from_json, rows_as_tuples


def test_k_lcf_around_anchor_reference_test_vectors(oracle):
    g = load_golden("lcs_anchor")
    assert len(g["from_reference_tests"]) >= 25
    for c in g["from_reference_tests"] + g["random"]:
        got = oracle.k_lcf_around_anchor(c["a"], c["b"], c["anchor_start"],
                                         c["anchor_end"], c["k"])
        assert list(got) == c["out"], c


def test_lcf_cover_reference_test_vectors(oracle):
    recs = load_golden("lcf_cover")
    assert len(recs) >= 15
    for c in recs:
        got = oracle.lcf_cover(c["probe_seq"], c["sequence"], c["kmer_start"],
                               c["kmer_end"], c["full_probe_len"],
                               c["full_sequence_len"], c["mismatches"],
                               c["lcf_thres"], c["island"])
        exp = None if c["out"] is None else tuple(c["out"])
        assert got == exp, c


def test_scan_reference_test_vectors(oracle):
    recs = load_golden("scan")
    assert len(recs) >= 20
    for c in recs:
        got = oracle.scan_sequence(c["sequence"], c["probes"],
                                   [tuple(e) for e in c["entries"]], c["k"],
                                   c["mismatches"], c["lcf_thres"], c["island"],
                                   c["merge"])
        exp = {int(k): [tuple(x) for x in v] for k, v in c["out"].items()}
        assert got == exp, (c["probes"], c["sequence"][:80])


def test_approx_multiuniverse_reference_test_vectors(oracle):
    recs = load_golden("setcover")
    assert len(recs) >= 28
    for c in recs:
        r = np.array(c["rows"], dtype=np.int64).reshape(-1, 4)
        got = oracle.approx_multiuniverse(r[:, 0], r[:, 1], r[:, 2], r[:, 3],
                                          c["num_sets"], c["num_universes"],
                                          c["costs"], c["universe_p"],
                                          c["ranks"])
        assert sorted(got) == c["out"], c


@pytest.mark.parametrize("name", ["scf_reference_tests", "scf_synthetic"])
def test_set_cover_filter_end_to_end(oracle, name):
    recs = load_golden(name)
    assert len(recs) >= 5
    for c in recs:
        if "np_random_state" in c:
            np.random.set_state(np_state_from_json(c["np_random_state"]))
        sel, inter = oracle.set_cover_filter(
            c["probes"], c["genomes"], c["mismatches"], c["lcf_thres"],
            c["island"], c["mismatches_tolerant"], c["lcf_thres_tolerant"],
            c["island_tolerant"], c["identify"], c["avoided_sequences"],
            c["coverage"], c["cover_extension"], c["kmer_probe_map_k"],
            return_intermediate=True)
        for gi, ids in enumerate(sel):
            got = sorted(c["probes"][gi][i] for i in ids)
            assert got == c["out"][gi], (name, c.get("name"), gi)
            if c.get("rows") is not None and len(c["probes"][gi]):
                exp_rows = sorted(tuple(x) for x in c["rows"][gi])
                assert rows_as_tuples(*inter[gi]["rows"]) == exp_rows
            if c.get("ranks") is not None and len(c["probes"][gi]):
                assert list(inter[gi]["ranks"]) == c["ranks"][gi]


def test_ndf_hamming(oracle):
    g = load_golden("ndf_hamming")
    recs = g["from_reference_tests"] + g["synthetic"]
    assert len(recs) >= 3
    for c in recs:
        assert oracle.lsh_num_tables(c["dist_thres"], c["dim"], c["k"],
                                     c["reporting_prob"]) == len(c["positions"])
        kept = oracle.ndf_hamming(c["probes"], c["dist_thres"], c["positions"])
        assert sorted(kept) == c["out"]


def test_ndf_minhash(oracle):
    """MinHash near-duplicate filter: vectors recorded from the reference run
    under PYTHONHASHSEED=0 (its MinHash family uses hash(str)); includes the
    known answers of the interpreter's str hash that the oracle restates."""
    import random
    g = load_golden("ndf_minhash")
    for x, h in g["str_hash"]:
        assert oracle.pyhash_seed0(x) == h
    recs = g["from_reference_tests"] + g["synthetic"]
    assert len(recs) >= 6
    for c in recs:
        assert oracle.minhash_num_tables(c["dist_thres"], c["k"],
                                         c["reporting_prob"]) == len(c["params"])
        kept = oracle.ndf_minhash(c["probes"], c["dist_thres"], c["params"],
                                  c["kmer_size"])
        assert sorted(kept) == c["out"]
    for c in g["synthetic"]:   # the draw order of (a, b)
        random.seed(c["seed"])
        got = oracle.minhash_draw_params(len(c["params"]), c["k"])
        assert [[list(ab) for ab in t] for t in got] == c["params"]


def test_pyhash_matches_this_interpreter(oracle):
    """Where the interpreter still hashes str with SipHash-2-4 (CPython <=
    3.10) the restatement must equal hash() under PYTHONHASHSEED=0."""
    import subprocess
    import sys
    if sys.version_info >= (3, 11):
        pytest.skip("CPython >= 3.11 hashes str with SipHash-1-3")
    strs = ["ACGTACGTAC", "NNNNNNNNNN", "GATTACAGAT", "C" * 37]
    out = subprocess.run(
        [sys.executable, "-c",
         "import sys; print([hash(x) for x in %r])" % (strs,)],
        env={"PYTHONHASHSEED": "0", "PATH": "/usr/bin:/bin"},
        capture_output=True, text=True, check=True).stdout
    assert eval(out) == [oracle.pyhash_seed0(x) for x in strs]


def test_scan_unmerged_and_coverage_analysis(oracle):
    """find_probe_covers_in_sequence(merge_overlapping=False) and the
    Analyzer's derived numbers (catch/coverage_analysis.py), recorded from the
    reference's own tests and seeded synthetic runs."""
    g = load_golden("coverage_analysis")
    assert len(g["scans_unmerged"]) >= 50
    for c in g["scans_unmerged"]:
        entries = [(int(a), int(b)) for a, b in c["entries"]]
        got = oracle.scan_sequence(c["sequence"], c["probes"], entries, c["k"],
                                   c["mismatches"], c["lcf_thres"], c["island"],
                                   merge=False)
        exp = {int(p): [tuple(x) for x in v] for p, v in c["out"].items()}
        assert got == exp
    recs = g["from_reference_tests"] + g["synthetic"]
    assert len(recs) >= 10
    for c in recs:
        if "np_seed" in c:
            np.random.seed(c["np_seed"])
        covers, bp, avg, counts = oracle.coverage_analysis(
            c["probes"], c["genomes"], c["mismatches"], c["lcf_thres"], 0,
            c["cover_extension"], c["kmer_probe_map_k"], c["rc_too"])
        assert [[[[list(x) for x in r] for r in j] for j in i] for i in covers] \
            == c["target_covers"]
        assert bp == c["bp_covered"]
        assert [[[list(r) for r in j] for j in i] for i in avg] == c["average_coverage"]
        assert counts == c["probe_map_counts"]


def test_merge_overlapping(oracle):
    assert oracle.merge_overlapping([(1, 5), (3, 7), (9, 12)]) == [(1, 7), (9, 12)]
    assert oracle.merge_overlapping([(1, 3), (3, 5)]) == [(1, 5)]
    assert oracle.merge_overlapping([]) == []
    assert oracle.merge_overlapping([(5, 6), (1, 2)]) == [(1, 2), (5, 6)]


def _dist_from_rows(rows):
    def dist(i, j):
        i, j = min(i, j), max(i, j)
        return rows[i][j - i - 1]
    return dist


def test_cluster_restatement_matches_reference(oracle):
    """Clustering pre-step (catch/utils/cluster.py, lsh.MinHashFamily with the
    md5 hash): signatures, connected components (including graphs where the
    early-stop heuristic makes the visiting order matter), hierarchical
    clusters, and whole clusterings, as recorded from the reference."""
    import random
    import sys
    g = load_golden("cluster")
    t = g["from_reference_tests"]
    assert len(t["cc"]) >= 4 and len(t["hier"]) >= 4 and len(t["minhash"]) >= 2
    same_python = g["python"].split(".")[:2] == sys.version.split()[0].split(".")[:2]
    for c in t["cc"]:
        got = oracle.find_connected_components(
            c["n"], _dist_from_rows(c["dist"]), c["threshold"],
            c.get("early_stop_threshold"))
        assert got == c["out"]
    for c in t["hier"]:
        got = oracle.cluster_hierarchically(np.asarray(c["dist_matrix"], dtype=np.float32),
                                            c["threshold"])
        assert got == c["out"]
    vals = (0.05, 0.3, 0.9)
    for c in g["cc_stress"]:
        n, cls = c["n"], c["classes"]
        rows, at = [], 0
        for i in range(n):
            rows.append([vals[int(x)] for x in cls[at:at + n - i - 1]])
            at += n - i - 1
        got = oracle.find_connected_components(n, _dist_from_rows(rows), c["threshold"],
                                               c["early_stop_threshold"])
        if same_python:      # the order of a set difference is the interpreter's
            assert got == c["out"]
        assert sorted(map(len, got), reverse=True) == [len(x) for x in got]
    for c in t["minhash"] + g["synthetic"]:
        sigs = [oracle.minhash_signature(s, c["k"], c["N"], c["a"], c["b"]) for s in c["seqs"]]
        assert [list(x) for x in sigs] == c["signatures"]
        if "seed" in c:
            random.seed(c["seed"])
            got = oracle.cluster_with_minhash_signatures(c["seqs"], c["k"], c["N"], c["threshold"],
                                                         c["method"])
            names = c["names"]
            assert [[names[i] for i in cl] for cl in got] == c["out"]
    assert any(len(s) - c["k"] + 1 < c["N"] for c in g["synthetic"] for s in c["seqs"])


def test_adapter_filter_restatement_matches_reference(oracle):
    """AdapterFilter votes and output (catch/filter/adapter_filter.py) as
    recorded from the reference under PYTHONHASHSEED=0; the restatement builds
    the k-mer map's sets from stand-ins hashed with the seed-0 string hash, so
    ties are broken as in that run whatever this process's hash seed is."""
    import sys
    g = load_golden("adapter_filter")
    same_python = g["python"].split(".")[:2] == sys.version.split()[0].split(".")[:2]
    recs = g["from_reference_tests"] + g["synthetic"]
    assert len(g["from_reference_tests"]) >= 4 and len(g["synthetic"]) >= 6
    exact = 0
    for c in recs:
        np.random.set_state(np_state_from_json(c["np_state"]))
        votes = oracle.adapter_votes(c["probes"], c["sequences"], c["mismatches"], c["lcf_thres"],
                                     c.get("island", 0), c["kmer_probe_map_k"],
                                     hash_fn=oracle.pyhash_seed0)
        if same_python:
            assert [list(v) for v in votes] == c["votes"]
            exact += 1
        # whatever the tie-breaks, every probe that hybridizes gets one vote per sequence
        assert [a + b for a, b in votes] == [a + b for a, b in c["votes"]]
        if "out" in c and same_python:
            np.random.set_state(np_state_from_json(c["np_state"]))
            a5, a3 = c["adapters"][0]
            b5, b3 = c["adapters"][1]
            assert oracle.adapter_filter(c["probes"], c["sequences"], (a5, a3), (b5, b3), c["mismatches"],
                                         c["lcf_thres"], c.get("island", 0), c["kmer_probe_map_k"],
                                         hash_fn=oracle.pyhash_seed0) == c["out"]
    assert exact == len(recs) or not same_python


def _random_instance(rng, nsets, nuniv, glen, partial, with_ranks):
    rows = []
    for s in range(nsets):
        for u in sorted(rng.choice(nuniv, size=rng.integers(1, nuniv + 1), replace=False)):
            iv, at = [], int(rng.integers(0, glen // 2))
            for _ in range(int(rng.integers(1, 4))):
                ln = int(rng.integers(1, 300))
                if at + ln > glen:
                    break
                iv.append((s, int(u), at, at + ln))
                at += ln + int(rng.integers(1, 200))
            rows += iv
    r = np.array(rows, dtype=np.int64).reshape(-1, 4)
    p = [float(rng.choice([1.0, 0.9, 0.5, 0.25, 0.0])) if partial else 1.0
         for _ in range(nuniv)]
    ranks = ([int(x) for x in rng.integers(0, 3, size=nsets)] if with_ranks
             else [0] * nsets)
    return r, p, ranks


def test_lazy_greedy_equals_restatement_on_reference_vectors(oracle):
    """The lazy evaluation makes the reference's picks in the reference's order
    on every unit-cost set cover instance recorded from its test suite."""
    n = 0
    for c in load_golden("setcover"):
        if (c["costs"] is not None and any(x != 1 for x in c["costs"])) or not c["rows"]:
            continue
        r = np.array(c["rows"], dtype=np.int64).reshape(-1, 4)
        glen = [int(r[r[:, 1] == u, 3].max()) if (r[:, 1] == u).any() else 0
                for u in range(c["num_universes"])]
        exp = oracle.approx_multiuniverse(r[:, 0], r[:, 1], r[:, 2], r[:, 3],
                                          c["num_sets"], c["num_universes"],
                                          c["costs"], c["universe_p"], c["ranks"])
        got = oracle.lazy_greedy(r[:, 0], r[:, 1], r[:, 2], r[:, 3],
                                 c["num_sets"], glen, c["universe_p"], c["ranks"])
        assert got == exp, c
        n += 1
    assert n >= 10


@pytest.mark.parametrize("partial,with_ranks", [(False, False), (True, False),
                                                (False, True), (True, True)])
def test_lazy_greedy_equals_restatement_random(oracle, partial, with_ranks):
    rng = np.random.Generator(np.random.PCG64(11 + 2 * partial + with_ranks))
    for it in range(40):
        nsets, nuniv = int(rng.integers(2, 60)), int(rng.integers(1, 6))
        glen = int(rng.integers(400, 3000))
        r, p, ranks = _random_instance(rng, nsets, nuniv, glen, partial, with_ranks)
        try:
            exp = oracle.approx_multiuniverse(r[:, 0], r[:, 1], r[:, 2], r[:, 3],
                                              nsets, nuniv, None, p, ranks)
        except IndexError:
            with pytest.raises(IndexError):
                oracle.lazy_greedy(r[:, 0], r[:, 1], r[:, 2], r[:, 3], nsets,
                                   [glen] * nuniv, p, ranks)
            continue
        got = oracle.lazy_greedy(r[:, 0], r[:, 1], r[:, 2], r[:, 3], nsets,
                                 [glen] * nuniv, p, ranks)
        assert got == exp, (it, partial, with_ranks)


def test_lazy_greedy_and_threaded_scan_on_a_pipeline(oracle):
    """Whole filter: threaded per-sequence scans + lazy greedy == the
    single-threaded restatement (picks in pick order)."""
    from tests.util import candidates, small_species
    genomes = small_species(seed=5, n=12, length=2500)
    cands = candidates(genomes, 100, 50)
    _, a = oracle.set_cover_filter([cands], [genomes], 2, 100,
                                   cover_extension=50, return_intermediate=True)
    oracle.set_threads(4)
    try:
        _, b = oracle.set_cover_filter([cands], [genomes], 2, 100,
                                       cover_extension=50, lazy=True,
                                       return_intermediate=True)
    finally:
        oracle.set_threads(1)
    assert rows_as_tuples(*a[0]["rows"]) == rows_as_tuples(*b[0]["rows"])
    assert a[0]["picks"] == b[0]["picks"] and len(a[0]["picks"]) > 5


def test_ndf_hamming_c_equals_python(oracle):
    """orc_ndf_hamming (C, for million-probe inputs) == oracle.ndf_hamming
    (pinned to the reference's vectors above) on the recorded cases and on
    near-duplicate-rich synthetic candidates."""
    import random
    from tests.util import candidates, small_species
    n = 0
    g = load_golden("ndf_hamming")
    for c in g["from_reference_tests"] + g["synthetic"]:
        kept = oracle.ndf_hamming_c(c["probes"], c["dist_thres"], c["positions"])
        assert kept == oracle.ndf_hamming(c["probes"], c["dist_thres"], c["positions"])
        assert sorted(kept) == c["out"]      # the reference's own answer
        n += 1
    for seed in (3, 4, 5):
        genomes = small_species(seed=seed, n=25, length=1500, d1=0.03, d2=0.01)
        strs = candidates(genomes, 100, 50, dedup=False)
        random.seed(seed)
        pos = oracle.lsh_draw_positions(oracle.lsh_num_tables(2, 100, 20), 20, 100)
        a = oracle.ndf_hamming(strs, 2, pos)
        oracle.set_threads(3)
        try:
            b = oracle.ndf_hamming_c(strs, 2, pos)
        finally:
            oracle.set_threads(1)
        assert a == b and len(a) < len(set(strs))
        n += 1
    assert n >= 8


def test_oracle_selects_what_the_live_reference_selected_on_indel_input(oracle):
    """S2i (S2's shape with insertions and deletions: a probe's hits in sibling
    strains are no longer at identical offsets, ranges touch and merge
    differently): the oracle's SetCoverFilter selection == the LIVE reference's
    (tests/golden/reference_runs.json, recorded by tools/time_reference.py)."""
    import hashlib
    import json
    from catch_amd.filter import candidate_probes
    from catch_amd.utils import synthetic
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_runs.json")) as f:
        run = [r for r in json.load(f)["runs"] if r["input"] == "S2i"][0]
    groups = synthetic.dataset("S2i", scale=run["scale"])
    cands = [list(dict.fromkeys(candidate_probes.candidate_strings_from_sequences(
        [s for g in grp for s in g], 100, 50))) for grp in groups]
    assert sum(map(len, cands)) == run["P"]
    oracle.set_threads(oracle.hw_threads())
    try:
        ids = oracle.set_cover_filter(cands, groups, 2, 100, coverage=1.0, cover_extension=50, lazy=True)
    finally:
        oracle.set_threads(1)
    sel = [sorted(c[i] for i in g) for c, g in zip(cands, ids)]
    assert sum(map(len, sel)) == run["probes_out"]
    assert hashlib.sha256("\n".join(",".join(g) for g in sel).encode()).hexdigest() == run["picks_sha256"]


def _real_runs():
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(here, "real_runs.json")) as f:
        d = json.load(f)
    return os.path.join(here, d["fasta"]), d["runs"]


def test_oracle_selects_what_the_live_reference_selected_on_real_ebola_genomes(oracle):
    """Real viral sequences (the first 30 records of the Ebola FASTA the reference's own
    tests hold: low-complexity runs, repeats, real indels, N runs -- what the synthetic
    genomes lack), four parameter sets incl. random anchors with truncated alignments
    (`-l 60`, np.random seeded) and partial coverage: the oracle's selection == the LIVE
    reference's (tests/golden/real_runs.json, recorded by tests/golden/make_real_golden.py).
    The 100-record runs are asserted on the GPU only (minutes of CPU here)."""
    import hashlib
    from catch_amd.filter import candidate_probes
    from catch_amd.utils import seq_io
    fasta, runs = _real_runs()
    genomes_all = [list(g.seqs) for g in seq_io.read_genomes_from_fasta(fasta)]
    assert len(genomes_all) == 100
    checked = 0
    oracle.set_threads(oracle.hw_threads())
    try:
        for r in runs:
            if r["records"] > 30:
                continue
            genomes = genomes_all[:r["records"]]
            pl = r["probe_length"]
            cands = list(dict.fromkeys(candidate_probes.candidate_strings_from_sequences(
                [s for g in genomes for s in g], pl, pl // 2)))
            assert len(cands) == r["P"] and sum(len(s) for g in genomes for s in g) == r["G"]
            if r["np_random_seed"] is not None:
                np.random.seed(r["np_random_seed"])
            ids = oracle.set_cover_filter([cands], [genomes], r["mismatches"], r["lcf_thres"], coverage=r["coverage"],
                                          cover_extension=r["cover_extension"], lazy=True)[0]
            sel = sorted(cands[i] for i in ids)
            assert len(sel) == r["probes_out"], r
            assert hashlib.sha256(",".join(sel).encode()).hexdigest() == r["picks_sha256"], r
            checked += 1
    finally:
        oracle.set_threads(1)
    assert checked >= 4


def _chain_runs():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ndf_scf_chains.json")) as f:
        return json.load(f)["runs"]


def _sha(strs):
    import hashlib
    return hashlib.sha256("\n".join(strs).encode()).hexdigest()


def test_oracle_ndf_then_scf_chains_equal_the_live_reference(oracle):
    """Near-duplicate filter, then set cover filter, recorded from the LIVE
    reference (tests/golden/make_chain_golden.py, PYTHONHASHSEED=0): the
    oracle's filter returns the kept probes in the reference's order --
    `list(to_include)`, the iteration order of a set of probes -- and its set
    cover over them selects the reference's probes (which of two equally good
    candidates wins depends on that order)."""
    import random
    from catch_amd.filter import candidate_probes
    from catch_amd.utils import synthetic
    runs = _chain_runs()
    assert len(runs) >= 4
    oracle.set_threads(oracle.hw_threads())
    try:
        for r in runs:
            genomes = synthetic.dataset(r["dataset"], scale=r["scale"])[r["group"]]
            cands = candidate_probes.candidate_strings_from_sequences([s for g in genomes for s in g], 100, 50)
            assert len(cands) == r["candidates"]
            random.seed(r["seed"])
            np.random.seed(r["seed"] + 1)
            if r["filter"] == "hamming":
                pos = oracle.lsh_draw_positions(oracle.lsh_num_tables(r["threshold"], 100, 20), 20, 100)
                kept = oracle.ndf_hamming_c(cands, r["threshold"], pos)
            else:
                params = oracle.minhash_draw_params(oracle.minhash_num_tables(r["threshold"]), 3)
                kept = oracle.ndf_minhash(cands, r["threshold"], params)
            assert len(kept) == r["kept"] and _sha(sorted(kept)) == r["kept_sorted_sha256"], r
            assert _sha(kept) == r["kept_in_order_sha256"], r          # the ORDER of list(set)
            ids = oracle.set_cover_filter([kept], [genomes], r["mismatches"], 100, coverage=1.0,
                                          cover_extension=50, lazy=True)[0]
            picks = sorted(kept[i] for i in ids)
            assert len(picks) == r["picks"] and _sha(picks) == r["picks_sorted_sha256"], r
    finally:
        oracle.set_threads(1)
