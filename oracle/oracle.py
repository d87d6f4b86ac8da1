"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY (never part of the product path).

ctypes front end to oracle/liboracle.so (the plain-C CPU restatement of the
broadinstitute/catch v1.5.2 hot path) plus the small pure-Python pieces of the
restatement (anchor-table construction, rank computation, near-duplicate
filter).  Only tests/, bench.py's `cpu_baseline` leg and
__graft_entry__.smoke() may import this module, and only as the checker.

Pinned against golden vectors recorded from the live reference
(tests/golden/*.json, generator tests/golden/make_golden.py); see
tests/test_oracle_golden.py.  Reference citations are relative to
/root/reference.
"""
import ctypes
import math
import os
import random
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """Compile liboracle.so with gcc (Makefile in this directory)."""
    src = os.path.join(_HERE, "catch_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_free.argtypes = [ctypes.c_void_p]
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_hw_threads.restype = ctypes.c_int
        L.orc_pyhash_seed0.argtypes = [_u8p, ctypes.c_int64]
        L.orc_pyhash_seed0.restype = ctypes.c_int64
        L.orc_minhash.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int64,
                                  ctypes.c_int64, ctypes.c_int64]
        L.orc_minhash.restype = ctypes.c_int64
        L.orc_k_lcf_around_anchor.argtypes = [
            _u8p, ctypes.c_int64, _u8p, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, ctypes.c_int64, _i64p, _i64p]
        L.orc_lcf_cover.argtypes = [
            _u8p, ctypes.c_int64, _u8p, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, ctypes.c_int64, _i64p, _i64p]
        L.orc_lcf_cover.restype = ctypes.c_int
        L.orc_merge_overlapping.argtypes = [_i64p, _i64p, ctypes.c_int64]
        L.orc_merge_overlapping.restype = ctypes.c_int64
        L.orc_scan_sequence.argtypes = [
            _u8p, ctypes.c_int64, _u8p, _i64p, _i32p, _i32p, ctypes.c_int64,
            ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int, ctypes.POINTER(_i32p), ctypes.POINTER(_i64p),
            ctypes.POINTER(_i64p)]
        L.orc_scan_sequence.restype = ctypes.c_int64
        L.orc_scan_first_seen.argtypes = [
            _u8p, ctypes.c_int64, _u8p, _i64p, _i32p, _i32p, ctypes.c_int64,
            ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            _i32p, ctypes.c_int64, _i64p, _i64p]
        L.orc_scan_first_seen.restype = None
        L.orc_make_sets.argtypes = [
            _u8p, _i64p, _i32p, ctypes.c_int64, _u8p, _i64p, _i32p, _i32p,
            ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(_i32p),
            ctypes.POINTER(_i32p), ctypes.POINTER(_i64p), ctypes.POINTER(_i64p)]
        L.orc_make_sets.restype = ctypes.c_int64
        L.orc_approx_multiuniverse.argtypes = [
            _i32p, _i32p, _i64p, _i64p, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, _f64p, _f64p, _i64p, _i64p]
        L.orc_approx_multiuniverse.restype = ctypes.c_int64
        L.orc_lazy_greedy.argtypes = [
            _i32p, _i32p, _i64p, _i64p, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, _i64p, _f64p, _i64p, _i64p]
        L.orc_lazy_greedy.restype = ctypes.c_int64
        L.orc_ndf_hamming.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int64, _i32p,
                                      ctypes.c_int64, ctypes.c_int64,
                                      ctypes.c_int64, _u8p]
        L.orc_ndf_hamming.restype = None
        _lib = L
    return _lib


def set_threads(n):
    """OpenMP threads for the per-sequence scans of make_sets (default 1).
    Returns the count in effect."""
    n = max(1, int(n))
    lib().orc_set_threads(n)
    return n


def hw_threads():
    return int(lib().orc_hw_threads())


def _bytes_arr(s):
    if isinstance(s, str):
        s = s.encode("latin-1")
    a = np.frombuffer(bytes(s), dtype=np.uint8)
    if a.size == 0:
        a = np.zeros(1, dtype=np.uint8)[:0]
    return np.ascontiguousarray(a)


def _p(a, t):
    return a.ctypes.data_as(t)


def _take(ptr, n, dtype):
    if n == 0:
        out = np.zeros(0, dtype=dtype)
    else:
        out = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
    lib().orc_free(ctypes.cast(ptr, ctypes.c_void_p))
    return out


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def k_lcf_around_anchor(a, b, anchor_start, anchor_end, k):
    """catch/utils/longest_common_substring.py:59-159 -> (length, start)."""
    aa, bb = _bytes_arr(a), _bytes_arr(b)
    ol, os_ = ctypes.c_int64(), ctypes.c_int64()
    lib().orc_k_lcf_around_anchor(_p(aa, _u8p), aa.size, _p(bb, _u8p), bb.size,
                                  anchor_start, anchor_end, k,
                                  ctypes.byref(ol), ctypes.byref(os_))
    return int(ol.value), int(os_.value)


def lcf_cover(probe_seq, sequence, kmer_start, kmer_end, full_probe_len,
              full_sequence_len, mismatches, lcf_thres, island=0):
    """catch/probe.py:1328-1344 -> (start, end) or None."""
    aa, bb = _bytes_arr(probe_seq), _bytes_arr(sequence)
    s, e = ctypes.c_int64(), ctypes.c_int64()
    ok = lib().orc_lcf_cover(_p(aa, _u8p), aa.size, _p(bb, _u8p), bb.size,
                             kmer_start, kmer_end, full_probe_len,
                             full_sequence_len, mismatches, lcf_thres, island,
                             ctypes.byref(s), ctypes.byref(e))
    return (int(s.value), int(e.value)) if ok else None


def merge_overlapping(intervals):
    """catch/utils/interval.py:288-316."""
    n = len(intervals)
    if n == 0:
        return []
    st = np.array([x[0] for x in intervals], dtype=np.int64)
    en = np.array([x[1] for x in intervals], dtype=np.int64)
    m = lib().orc_merge_overlapping(_p(st, _i64p), _p(en, _i64p), n)
    return [(int(st[i]), int(en[i])) for i in range(m)]


# --------------------------------------------------------------------------
# anchor (k-mer seed) table: catch/probe.py:356-577
# --------------------------------------------------------------------------
def pigeonhole_k(probe_length, mismatches, min_k):
    """catch/probe.py:470-494; returns k or None when k < min_k."""
    if mismatches == 0:
        k = probe_length
    else:
        k = int(probe_length / mismatches)
        if k == float(probe_length) / mismatches:
            k -= 1
        while probe_length % k != 0:
            k -= 1
    if k < min_k:
        return None
    return k


def anchor_table(probe_strs, mismatches, lcf_thres, min_k=20, k=20,
                 num_kmers_per_probe=20, with_draws=False):
    """construct_kmer_probe_map_to_find_probe_covers (catch/probe.py:507-577).
    probe_strs may contain duplicates (the reference iterates every Probe
    object, so in random mode np.random is consumed once per input probe,
    :391-401, and equal probes pool their positions because Probe hashes by
    sequence).  Returns (k, entries): entries = sorted unique
    (unique_probe_index, position) pairs, unique indices in first-seen order
    (see _unique_last).  with_draws: also the (input index, position) pairs in
    the order the reference adds them to its k-mer map."""
    if len(probe_strs) == 0:
        return (None, [], []) if with_draws else (None, [])
    uidx = {}
    for p in probe_strs:
        uidx.setdefault(p, len(uidx))
    L = len(probe_strs[0])
    differ = any(len(p) != L for p in probe_strs)
    use_random = (mismatches is None or lcf_thres is None or differ
                  or lcf_thres < L)
    kk = None
    if not use_random:
        kk = pigeonhole_k(L, mismatches, min_k)
        if kk is None:
            use_random = True
    entries = set()
    draws = []
    if use_random:
        for i, p in enumerate(probe_strs):
            if k > len(p):
                raise ValueError("k is larger than the length of a probe")
            n_kmers = len(p) - k + 1
            for pos in np.random.choice(n_kmers, size=num_kmers_per_probe,
                                        replace=True):
                entries.add((uidx[p], int(pos)))
                draws.append((i, int(pos)))
        kk = k
    else:
        for i, p in enumerate(probe_strs):
            for pos in range(0, L, kk):
                entries.add((uidx[p], pos))
                draws.append((i, pos))
    if with_draws:
        return kk, sorted(entries), draws
    return kk, sorted(entries)


def _pack_probes(probe_strs):
    off = np.zeros(len(probe_strs) + 1, dtype=np.int64)
    for i, p in enumerate(probe_strs):
        off[i + 1] = off[i] + len(p)
    buf = _bytes_arr("".join(probe_strs))
    return buf, off


def _pack_entries(entries):
    ep = np.array([e[0] for e in entries], dtype=np.int32)
    eo = np.array([e[1] for e in entries], dtype=np.int32)
    return ep, eo


# --------------------------------------------------------------------------
# scan: catch/probe.py:1008-1271
# --------------------------------------------------------------------------
def scan_sequence(sequence, probe_strs, entries, k, mismatches, lcf_thres,
                  island=0, merge=True):
    """find_probe_covers_in_sequence -> {probe_index: [(start, end), ...]}."""
    seq = _bytes_arr(sequence)
    buf, off = _pack_probes(probe_strs)
    ep, eo = _pack_entries(entries)
    op, os_, oe = _i32p(), _i64p(), _i64p()
    n = lib().orc_scan_sequence(_p(seq, _u8p), seq.size, _p(buf, _u8p),
                                _p(off, _i64p), _p(ep, _i32p), _p(eo, _i32p),
                                len(entries), k, mismatches, lcf_thres, island,
                                1 if merge else 0, ctypes.byref(op),
                                ctypes.byref(os_), ctypes.byref(oe))
    pr = _take(op, n, np.int32)
    st = _take(os_, n, np.int64)
    en = _take(oe, n, np.int64)
    out = {}
    for i in range(n):
        out.setdefault(int(pr[i]), []).append((int(st[i]), int(en[i])))
    return out


def make_sets(probe_strs, entries, k, genomes, mismatches, lcf_thres,
              island=0, cover_extension=0):
    """SetCoverFilter._make_sets (catch/filter/set_cover_filter.py:359-470).
    genomes: list of genomes, each a list of sequence strings.
    Returns int arrays (probe, universe, start, end) sorted by
    (probe, universe, start), IntervalSet-normalised."""
    seqs, seq_genome = [], []
    for j, g in enumerate(genomes):
        for s in g:
            seqs.append(s)
            seq_genome.append(j)
    soff = np.zeros(len(seqs) + 1, dtype=np.int64)
    for i, s in enumerate(seqs):
        soff[i + 1] = soff[i] + len(s)
    sbuf = _bytes_arr("".join(seqs))
    sg = np.array(seq_genome, dtype=np.int32)
    buf, off = _pack_probes(probe_strs)
    ep, eo = _pack_entries(entries)
    op, ou, os_, oe = _i32p(), _i32p(), _i64p(), _i64p()
    n = lib().orc_make_sets(_p(sbuf, _u8p), _p(soff, _i64p), _p(sg, _i32p),
                            len(seqs), _p(buf, _u8p), _p(off, _i64p),
                            _p(ep, _i32p), _p(eo, _i32p), len(entries), k,
                            mismatches, lcf_thres, island, cover_extension,
                            ctypes.byref(op), ctypes.byref(ou),
                            ctypes.byref(os_), ctypes.byref(oe))
    return (_take(op, n, np.int32), _take(ou, n, np.int32),
            _take(os_, n, np.int64), _take(oe, n, np.int64))


# --------------------------------------------------------------------------
# greedy multi-universe set cover: catch/utils/set_cover.py:147-615
# --------------------------------------------------------------------------
def approx_multiuniverse(row_set, row_univ, row_start, row_end, num_sets,
                         num_universes, costs=None, universe_p=None,
                         ranks=None):
    """Rows sorted by (set, universe, start), normalised per (set, universe).
    Returns the picked set ids in pick order."""
    rs = np.ascontiguousarray(row_set, dtype=np.int32)
    ru = np.ascontiguousarray(row_univ, dtype=np.int32)
    st = np.ascontiguousarray(row_start, dtype=np.int64)
    en = np.ascontiguousarray(row_end, dtype=np.int64)
    P, U = int(num_sets), int(num_universes)
    c = (np.ones(P, dtype=np.float64) if costs is None
         else np.ascontiguousarray(costs, dtype=np.float64))
    up = (np.ones(U, dtype=np.float64) if universe_p is None
          else np.ascontiguousarray(universe_p, dtype=np.float64))
    rk = (np.ones(P, dtype=np.int64) if ranks is None
          else np.ascontiguousarray(ranks, dtype=np.int64))
    out = np.zeros(max(P, 1), dtype=np.int64)
    n = lib().orc_approx_multiuniverse(_p(rs, _i32p), _p(ru, _i32p),
                                       _p(st, _i64p), _p(en, _i64p), rs.size,
                                       P, U, _p(c, _f64p), _p(up, _f64p),
                                       _p(rk, _i64p), _p(out, _i64p))
    if n < 0:
        raise IndexError("rank list exhausted (reference raises IndexError)")
    return [int(x) for x in out[:n]]


def lazy_greedy(row_set, row_univ, row_start, row_end, num_sets, genome_len,
                universe_p=None, ranks=None):
    """orc_lazy_greedy: the same picks in the same order as
    approx_multiuniverse with unit costs, by lazy evaluation over bitmaps (see
    the C source).  genome_len[u] = length of universe u's coordinate space."""
    rs = np.ascontiguousarray(row_set, dtype=np.int32)
    ru = np.ascontiguousarray(row_univ, dtype=np.int32)
    st = np.ascontiguousarray(row_start, dtype=np.int64)
    en = np.ascontiguousarray(row_end, dtype=np.int64)
    gl = np.ascontiguousarray(genome_len, dtype=np.int64)
    P, U = int(num_sets), int(gl.size)
    up = (np.ones(U, dtype=np.float64) if universe_p is None
          else np.ascontiguousarray(universe_p, dtype=np.float64))
    rk = (np.ones(P, dtype=np.int64) if ranks is None
          else np.ascontiguousarray(ranks, dtype=np.int64))
    out = np.zeros(max(P, 1), dtype=np.int64)
    n = lib().orc_lazy_greedy(_p(rs, _i32p), _p(ru, _i32p), _p(st, _i64p),
                              _p(en, _i64p), rs.size, P, U, _p(gl, _i64p),
                              _p(up, _f64p), _p(rk, _i64p), _p(out, _i64p))
    if n < 0:
        raise IndexError("rank list exhausted (reference raises IndexError)")
    return out[:n].tolist()


# --------------------------------------------------------------------------
# SetCoverFilter end to end: catch/filter/set_cover_filter.py:794-930
# --------------------------------------------------------------------------
_RC = {ord('A'): 'T', ord('T'): 'A', ord('C'): 'G', ord('G'): 'C'}


def reverse_complement(s):
    """catch/filter/set_cover_filter.py:515-521 (non-ACGT map to themselves)."""
    return "".join(_RC.get(ord(b), b) for b in s[::-1])


def _unique_last(probe_strs):
    """Unique probe strings in first-seen order; owner index = LAST index with
    that string (catch/filter/set_cover_filter.py:408-412)."""
    owner = {}
    for i, p in enumerate(probe_strs):
        owner[p] = i
    uniq = list(owner.keys())
    return uniq, [owner[p] for p in uniq]


def tolerant_bp(uniq, entries, k, sequence, mismatches, lcf_thres, island,
                rc_too=True):
    """_compute_tolerant_bp_covered_within_sequence (:472-529)."""
    out = {}
    seqs = [sequence] + ([reverse_complement(sequence)] if rc_too else [])
    for s in seqs:
        for pi, ranges in scan_sequence(s, uniq, entries, k, mismatches,
                                        lcf_thres, island, True).items():
            out[pi] = out.get(pi, 0) + sum(e - st for st, e in ranges)
    return out


def make_ranks(probe_strs, genomes_grouped, params, avoided_sequences):
    """_make_ranks (:614-735). avoided_sequences: list of sequence strings in
    the order seq_io.iterate_fasta yields them over all avoided FASTA files."""
    identify = params.get("identify", False)
    uniq, _ = _unique_last(probe_strs)
    uidx = {p: i for i, p in enumerate(uniq)}
    need = identify or len(avoided_sequences) > 0
    if need:
        m_t, l_t, i_t = (params["mismatches_tolerant"],
                         params["lcf_thres_tolerant"],
                         params["island_tolerant"])
        kk = params.get("kmer_probe_map_k", 20)
        k, entries = anchor_table(probe_strs, m_t, l_t, min_k=kk, k=kk)
    rank_val = {}
    if identify:
        hits = [0] * len(uniq)
        for group in genomes_grouped:
            tot = {}
            for g in group:
                for s in g:
                    for pi, bp in tolerant_bp(uniq, entries, k, s, m_t, l_t,
                                              i_t, True).items():
                        tot[pi] = tot.get(pi, 0) + bp
            for pi, bp in tot.items():
                if bp >= 1:
                    hits[pi] += 1
        for pi in range(len(uniq)):
            rank_val[pi] = (0, hits[pi])
    else:
        for pi in range(len(uniq)):
            rank_val[pi] = (0, 0)
    tot = {}
    for s in avoided_sequences:
        for pi, bp in tolerant_bp(uniq, entries, k, s, m_t, l_t, i_t,
                                  True).items():
            tot[pi] = tot.get(pi, 0) + bp
    for pi, bp in tot.items():
        if bp > 0:
            rank_val[pi] = (1, bp)
    all_t = sorted(set(rank_val.values()))
    tidx = {t: i for i, t in enumerate(all_t)}
    return [tidx[rank_val[uidx[p]]] for p in probe_strs]


def universe_p(coverage, genome_sizes):
    """_make_universe_p (:761-792)."""
    if coverage <= 1.0:
        return [coverage for _ in genome_sizes]
    return [float(min(coverage, sz)) / sz for sz in genome_sizes]


def set_cover_filter(probes_grouped, genomes_grouped, mismatches, lcf_thres,
                     island=0, mismatches_tolerant=None,
                     lcf_thres_tolerant=None, island_tolerant=None,
                     identify=False, avoided_sequences=(), coverage=1.0,
                     cover_extension=0, kmer_probe_map_k=20,
                     return_intermediate=False, lazy=False):
    """SetCoverFilter.__init__ + _filter (catch/filter/set_cover_filter.py
    :199-357, :902-930).  probes_grouped: list of lists of probe strings;
    genomes_grouped: list (per group) of genomes, each a list of sequence
    strings.  Returns per-group sorted lists of selected probe indices."""
    if not mismatches_tolerant:
        mismatches_tolerant = mismatches
    if not lcf_thres_tolerant:
        lcf_thres_tolerant = lcf_thres
    if not island_tolerant:
        island_tolerant = island
    params = dict(identify=identify, mismatches_tolerant=mismatches_tolerant,
                  lcf_thres_tolerant=lcf_thres_tolerant,
                  island_tolerant=island_tolerant,
                  kmer_probe_map_k=kmer_probe_map_k)
    selected, inter = [], []
    for gi, (probe_strs, genomes) in enumerate(zip(probes_grouped,
                                                   genomes_grouped)):
        probe_strs = list(probe_strs)
        P = len(probe_strs)
        if P == 0:
            rows = (np.zeros(0, np.int32), np.zeros(0, np.int32),
                    np.zeros(0, np.int64), np.zeros(0, np.int64))
            k, entries = None, []
        else:
            uniq, owner = _unique_last(probe_strs)
            k, entries = anchor_table(probe_strs, mismatches, lcf_thres,
                                      min_k=kmer_probe_map_k,
                                      k=kmer_probe_map_k)
            up_, uu, us, ue = make_sets(uniq, entries, k, genomes, mismatches,
                                        lcf_thres, island, cover_extension)
            own = np.array(owner, dtype=np.int32)
            rs = own[up_] if up_.size else up_
            order = np.lexsort((us, uu, rs))
            rows = (rs[order], uu[order], us[order], ue[order])
        ranks = make_ranks(probe_strs, genomes_grouped, params,
                           list(avoided_sequences)) if P else []
        up = universe_p(coverage, [sum(len(s) for s in g) for g in genomes])
        if lazy and P:
            picks = lazy_greedy(rows[0], rows[1], rows[2], rows[3], P,
                                [sum(len(s) for s in g) for g in genomes], up,
                                ranks)
        else:
            picks = approx_multiuniverse(rows[0], rows[1], rows[2], rows[3], P,
                                         len(genomes), None, up,
                                         ranks) if P else []
        selected.append(sorted(picks))
        inter.append(dict(k=k, entries=entries, rows=rows, ranks=ranks,
                          universe_p=up, picks=picks))
    if return_intermediate:
        return selected, inter
    return selected


# --------------------------------------------------------------------------
# near-duplicate filter (Hamming): catch/filter/near_duplicate_filter.py
# :47-142 + catch/utils/lsh.py:16-45, :218-320
# --------------------------------------------------------------------------
def lsh_num_tables(dist_thres, dim, k, reporting_prob=0.80):
    """catch/utils/lsh.py:268-276."""
    P1 = 1.0 - float(dist_thres) / float(dim)
    if P1 == 1.0:
        return 1
    return int(math.ceil(math.log(1.0 - reporting_prob,
                                  1.0 - math.pow(P1, k))))


def lsh_draw_positions(num_tables, k, dim):
    """Positions in the order the reference draws them: per table, k calls of
    random.randint(0, dim-1) (lsh.py:284-287 -> :224 -> :28)."""
    return [[random.randint(0, dim - 1) for _ in range(k)]
            for _ in range(num_tables)]


def as_list_of_set(kept):
    """`list(to_include)` (near_duplicate_filter.py:103): the kept probes in the
    iteration order of the Python SET they were added to in inclusion order.
    Probe objects hash by hash(seq_str) (catch/probe.py:324-329); a real set of
    stand-ins with the PYTHONHASHSEED=0 hashes (pyhash_seed0) gives the order
    the reference produces under that setting, in any interpreter."""
    include = set()
    for p in kept:
        include.add(_HashedProbe(p, pyhash_seed0(p)))
    return [h.s for h in include]


def ndf_hamming(probe_strs, dist_thres, positions, inclusion_order=False):
    """NearDuplicateFilter._filter with the Hamming family; `positions` is the
    per-table list of sampled positions.  Returns the kept probe strings as
    the reference does, `list(to_include)` (as_list_of_set); inclusion_order:
    in the order they were included instead (= multiplicity-sorted order)."""
    occ = {}
    for p in probe_strs:
        occ[p] = occ.get(p, 0) + 1
    # stable sort by count desc (near_duplicate_filter.py:64-66)
    order = [p for p, _ in sorted(occ.items(), key=lambda kv: kv[1],
                                  reverse=True)]
    tables = []
    for pos in positions:
        ht = {}
        for p in occ.keys():
            ht.setdefault(tuple(p[i] for i in pos), []).append(p)
        tables.append(ht)
    arr = {p: np.frombuffer(p.encode("latin-1"), dtype=np.uint8)
           for p in occ.keys()}
    include, exclude, kept = set(), set(), []
    for p in order:
        if p in exclude:
            continue
        include.add(p)
        kept.append(p)
        for pos, ht in zip(positions, tables):
            for q in ht[tuple(p[i] for i in pos)]:
                if q in include:
                    continue
                if len(q) != len(p):
                    raise ValueError("Sequences must be of same length")
                if int(np.count_nonzero(arr[p] != arr[q])) <= dist_thres:
                    exclude.add(q)
    return kept if inclusion_order else as_list_of_set(kept)


def ndf_hamming_c(probe_strs, dist_thres, positions, inclusion_order=False):
    """ndf_hamming with the bucket search and the sequential pass in C
    (orc_ndf_hamming), for inputs of millions of probes.  Same result."""
    occ = {}
    for p in probe_strs:
        occ[p] = occ.get(p, 0) + 1
    order = [p for p, _ in sorted(occ.items(), key=lambda kv: kv[1],
                                  reverse=True)]
    if not order:
        return []
    L = len(order[0])
    if any(len(p) != L for p in order):
        raise ValueError("Sequences must be of same length")
    buf = _bytes_arr("".join(order))
    pos = np.ascontiguousarray(positions, dtype=np.int32)
    keep = np.zeros(len(order), dtype=np.uint8)
    lib().orc_ndf_hamming(_p(buf, _u8p), len(order), L, _p(pos, _i32p),
                          pos.shape[0], pos.shape[1], int(dist_thres),
                          _p(keep, _u8p))
    kept = [p for p, k in zip(order, keep) if k]
    return kept if inclusion_order else as_list_of_set(kept)


# --------------------------------------------------------------------------
# near-duplicate filter, MinHash family: catch/utils/lsh.py:48-215,
# catch/filter/near_duplicate_filter.py:148-190
# --------------------------------------------------------------------------
MINHASH_P = 2 ** 31 - 1


def pyhash_seed0(s):
    """hash(s) of CPython 3.4-3.10 for an ASCII str under PYTHONHASHSEED=0
    (see catch_oracle.c: the reference's MinHash uses abs(hash(kmer)))."""
    b = _bytes_arr(s)
    return int(lib().orc_pyhash_seed0(_p(b, _u8p), len(s)))


def minhash_num_tables(dist_thres, k=3, reporting_prob=0.80):
    """lsh.py:268-276 with MinHashFamily.P1 (:160-174) = 1 - dist."""
    P1 = 1.0 - dist_thres
    if P1 == 1.0:
        return 1
    return int(math.ceil(math.log(1.0 - reporting_prob,
                                  1.0 - math.pow(P1, k))))


def minhash_draw_params(num_tables, k):
    """(a, b) of every hash function in the order the reference draws them:
    per table, k times make_h(): a = random.randint(1, p), b =
    random.randint(0, p) (lsh.py:284-287 -> :224 -> :95-96)."""
    return [[(random.randint(1, MINHASH_P), random.randint(0, MINHASH_P))
             for _ in range(k)] for _ in range(num_tables)]


def jaccard_dist(a, b, kmer_size):
    """near_duplicate_filter.py:148-157."""
    ak = set(a[i:i + kmer_size] for i in range(len(a) - kmer_size + 1))
    bk = set(b[i:i + kmer_size] for i in range(len(b) - kmer_size + 1))
    return 1.0 - float(len(ak & bk)) / len(ak | bk)


def ndf_minhash(probe_strs, dist_thres, params, kmer_size=10, inclusion_order=False):
    """NearDuplicateFilter._filter with the MinHash family (N = 1, hash(str)
    under PYTHONHASHSEED=0); `params` = minhash_draw_params(...).  Returns the
    kept probe strings as the reference does (as_list_of_set)."""
    occ = {}
    for p in probe_strs:
        occ[p] = occ.get(p, 0) + 1
    order = [p for p, _ in sorted(occ.items(), key=lambda kv: kv[1],
                                  reverse=True)]
    L = lib()

    def g(p, fns):
        b = _bytes_arr(p)
        return tuple(int(L.orc_minhash(_p(b, _u8p), len(p), kmer_size, a_, b_))
                     for a_, b_ in fns)
    keys = {p: [g(p, fns) for fns in params] for p in occ.keys()}
    tables = []
    for t in range(len(params)):
        ht = {}
        for p in occ.keys():
            ht.setdefault(keys[p][t], []).append(p)
        tables.append(ht)
    include, exclude, kept = set(), set(), []
    for p in order:
        if p in exclude:
            continue
        include.add(p)
        kept.append(p)
        for t, ht in enumerate(tables):
            for q in ht[keys[p][t]]:
                if q in include:
                    continue
                if jaccard_dist(p, q, kmer_size) <= dist_thres:
                    exclude.add(q)
    return kept if inclusion_order else as_list_of_set(kept)


# --------------------------------------------------------------------------
# coverage analysis: catch/coverage_analysis.py:183-335
# --------------------------------------------------------------------------
def coverage_analysis(probe_strs, genomes_grouped, mismatches, lcf_thres,
                      island=0, cover_extension=0, kmer_probe_map_k=10,
                      rc_too=True):
    """Analyzer._find_covers_in_target_genomes + _compute_bp_covered... +
    _compute_average_coverage... .  genomes_grouped[i][j] = list of sequence
    strings.  Returns (target_covers, bp_covered, average_coverage,
    probe_map_counts) with target_covers[i][j][r] a sorted list of (start, end)
    (r = 0 forward, 1 reverse complement), average_coverage[i][j][r] =
    (over all bases, over unambiguous bases), probe_map_counts per probe."""
    k, entries = anchor_table(probe_strs, mismatches, lcf_thres,
                              min_k=kmer_probe_map_k, k=kmer_probe_map_k)
    uniq, owner = _unique_last(probe_strs)
    counts = [0] * len(probe_strs)
    covers, bp, avg = [], [], []
    for grp in genomes_grouped:
        cg, bg, ag = [], [], []
        for gnm in grp:
            cj, bj, aj = [], [], []
            size_all = sum(len(s) for s in gnm)
            size_unambig = sum(s.count(b) for s in gnm for b in "ATCG")
            for rc in ((False, True) if rc_too else (False,)):
                out, so_far = [], 0
                for seq in gnm:
                    s = reverse_complement(seq) if rc else seq
                    cov = scan_sequence(s, uniq, entries, k, mismatches,
                                        lcf_thres, island, merge=False)
                    for p, ranges in cov.items():
                        if not rc:
                            counts[owner[p]] += 1
                        for a, b in ranges:
                            out.append((max(0, a - cover_extension) + so_far,
                                        min(len(s), b + cover_extension) + so_far))
                    so_far += len(s)
                out.sort()
                cj.append(out)
                bj.append(sum(b - a for a, b in merge_overlapping(out)))
                total = sum(b - a for a, b in out)
                aj.append((float(total) / size_all, float(total) / size_unambig))
            cg.append(cj)
            bg.append(bj)
            ag.append(aj)
        covers.append(cg)
        bp.append(bg)
        avg.append(ag)
    return covers, bp, avg, counts


# --------------------------------------------------------------------------
# clustering pre-step: catch/utils/cluster.py + lsh.MinHashFamily (md5 hash)
# --------------------------------------------------------------------------
def md5_kmer_hash(kmer, a, b):
    """lsh.py:106-111: (a * int(md5(kmer).hexdigest(), 16) + b) mod (2^31-1)."""
    import hashlib
    x = int(hashlib.md5(kmer.encode("utf-8")).hexdigest(), 16)
    return (a * x + b) % MINHASH_P


def minhash_signature(s, kmer_size, N, a, b):
    """MinHashFamily.make_h()'s h(s) (lsh.py:113-153): the N smallest k-mer
    hashes (with multiplicity), ascending; when the sequence has fewer than N
    k-mers they are taken in whole extra rounds."""
    assert kmer_size <= len(s)
    num_kmers = len(s) - kmer_size + 1
    vals = [md5_kmer_hash(s[i:i + kmer_size], a, b) for i in range(num_kmers)]
    rounds = 1
    while rounds * num_kmers < N:
        rounds += 1
    return tuple(sorted(vals * rounds)[:N])


def signature_common(hA, hB, N):
    """The walk of MinHashFamily.estimate_jaccard_dist (lsh.py:190-210):
    returns (intersect_count, union_count)."""
    i = j = inter = union = 0
    while i < len(hA) and j < len(hB):
        if union == N:
            break
        if hA[i] < hB[j]:
            i += 1
        elif hA[i] > hB[j]:
            j += 1
        else:
            inter += 1
            i += 1
            j += 1
        union += 1
    return inter, union


def estimate_jaccard_dist(hA, hB, N):
    inter, union = signature_common(hA, hB, N)
    return 1.0 - float(inter) / union


def jaccard_dist_from_mash_dist(mash_dist, k):
    """cluster.py:47-68 (Mash eq. 4 solved for j)."""
    return 1.0 - 1.0 / (2.0 * np.exp(k * mash_dist) - 1)


def condensed_dist_matrix(n, dist_fn):
    """cluster.py:102-194: float32 (c_float) condensed matrix, SciPy order."""
    out = np.zeros(n * (n - 1) // 2, dtype=np.float32)
    idx = 0
    for i in range(n):
        for j in range(i + 1, n):
            out[idx] = dist_fn(i, j)
            idx += 1
    return out


def cluster_hierarchically(dist_matrix, threshold):
    """cluster.py:197-232: average linkage, cut at `threshold`, clusters by
    descending size (ties keep cluster-number order)."""
    from scipy.cluster import hierarchy
    if len(dist_matrix) == 0:
        return [[0]]
    link = hierarchy.linkage(dist_matrix, method="average")
    labels = hierarchy.fcluster(link, threshold, criterion="distance")
    members = {}
    for i, c in enumerate(labels):
        members.setdefault(int(c), []).append(i)
    order = sorted(range(min(members), max(members) + 1),
                   key=lambda c: -len(members[c]))
    return [members[c] for c in order]


def find_connected_components(n, dist_fn, threshold, early_stop_threshold=None):
    """cluster.py:235-355.  Depth-first search with the early-stop heuristic:
    a neighbour within `early_stop_threshold` is absorbed without being
    explored.  The outcome can depend on the order in which neighbours are
    examined, which in the reference is the iteration order of a Python set
    difference; the same set operations are performed here so the order is the
    interpreter's own."""
    if early_stop_threshold is None:
        early_stop_threshold = jaccard_dist_from_mash_dist(0.02, 12)
    remaining = set(range(n))
    done = set()
    comps = []
    for i in range(n):
        if i in done:
            continue
        seen = set()
        stack = [i]
        queued = {i}
        while len(stack) > 0:
            j = stack.pop()
            if j in seen:
                continue
            seen.add(j)
            for k in list(remaining - queued):
                d = dist_fn(j, k)
                if d <= threshold:
                    if d <= early_stop_threshold:
                        seen.add(k)
                    else:
                        stack.append(k)
                    queued.add(k)
        done.update(seen)
        remaining -= seen
        comps.append(sorted(seen))
    comps.sort(key=len, reverse=True)
    return comps


def cluster_with_minhash_signatures(seqs, k=12, N=100, threshold=0.1,
                                    cluster_method="simple"):
    """cluster.py:358-430 on a list of sequences (indices stand for the
    reference's dict keys).  Consumes random.randint twice (a, b)."""
    a = random.randint(1, MINHASH_P)
    b = random.randint(0, MINHASH_P)
    sigs = [minhash_signature(s, k, N, a, b) for s in seqs]
    thr = jaccard_dist_from_mash_dist(threshold, k)

    def dist(i, j):
        return estimate_jaccard_dist(sigs[i], sigs[j], N)
    if cluster_method == "simple":
        return find_connected_components(len(seqs), dist, thr)
    if cluster_method == "hierarchical":
        return cluster_hierarchically(condensed_dist_matrix(len(seqs), dist),
                                      thr)
    raise ValueError("Unknown cluster_method '%s'" % cluster_method)


def fragments_of(seq, fragment_length):
    """genome.py:64-100 with include_full_end=True."""
    out = []
    for i in range(0, len(seq), fragment_length):
        f = seq[i:i + fragment_length]
        if len(f) < fragment_length:
            f = seq[max(0, len(seq) - fragment_length):]
        out.append(f)
    return out


# --------------------------------------------------------------------------
# adapter filter: catch/filter/adapter_filter.py:191-392
# --------------------------------------------------------------------------
class _HashedProbe:
    """Stands for a reference Probe inside the k-mer map's sets: equal by
    sequence, hash = hash_fn(sequence) (the reference's is hash(seq_str),
    catch/probe.py:324-329; pyhash_seed0 reproduces it for PYTHONHASHSEED=0)."""
    __slots__ = ("s", "h")

    def __init__(self, s, h):
        self.s, self.h = s, h

    def __hash__(self):
        return self.h

    def __eq__(self, other):
        return self.s == other.s


def kmer_entry_ranks(probe_strs, entries, draws, k, hash_fn=None):
    """Rank of every anchor entry inside its k-mer's entry list.  The
    reference lists a k-mer's entries by iterating the Python set of
    (Probe, pos) tuples built in `draws` order (probe.py:393-401 / :496-503,
    SharedKmerProbeMap.construct :739-747); the same sets are built here from
    stand-in objects with the same hashes."""
    hash_fn = hash_fn or hash
    uidx = {}
    for p in probe_strs:
        uidx.setdefault(p, len(uidx))
    objs = [_HashedProbe(p, hash_fn(p)) for p in probe_strs]
    kmap = {}
    for i, pos in draws:
        kmap.setdefault(probe_strs[i][pos:pos + k], set()).add((objs[i], pos))
    rank = {}
    for kmer, members in kmap.items():
        for r, (o, pos) in enumerate(members):
            rank[(uidx[o.s], pos)] = r
    return [rank[e] for e in entries]


def scan_first_seen(sequence, uniq, entries, k, mismatches, lcf_thres, island,
                    ent_rank):
    """{probe: (first accepted k-mer position, its entry)} for the probes that
    hybridize somewhere in `sequence`."""
    seq = _bytes_arr(sequence)
    buf, off = _pack_probes(uniq)
    ep, eo = _pack_entries(entries)
    er = np.asarray(ent_rank, dtype=np.int32)
    fi = np.zeros(max(len(uniq), 1), dtype=np.int64)
    fe = np.zeros(max(len(uniq), 1), dtype=np.int64)
    lib().orc_scan_first_seen(_p(seq, _u8p), seq.size, _p(buf, _u8p), _p(off, _i64p),
                              _p(ep, _i32p), _p(eo, _i32p), len(entries), k,
                              mismatches, lcf_thres, island, _p(er, _i32p),
                              len(uniq), _p(fi, _i64p), _p(fe, _i64p))
    return {p: (int(fi[p]), int(fe[p])) for p in range(len(uniq)) if fi[p] >= 0}


def schedule(intervals):
    """interval.schedule (catch/utils/interval.py:319-358): stable sort by end,
    take every interval starting at or after the last chosen end."""
    chosen, last_end = [], None
    for (start, end), obj in sorted(intervals, key=lambda x: x[0][1]):
        if last_end is None or start >= last_end:
            chosen.append(obj)
            last_end = end
    return chosen


def adapter_votes(probe_strs, sequences, mismatches, lcf_thres, island=0,
                  kmer_probe_map_k=20, hash_fn=None, detail=None):
    """AdapterFilter._make_votes_across_target_genomes (:299-361) over
    `sequences` (all sequences of all genomes of all groups, in order).
    Returns [(A votes, B votes)] per input probe.  Consumes np.random like the
    reference when the anchors are random."""
    k, entries, draws = anchor_table(probe_strs, mismatches, lcf_thres,
                                     min_k=kmer_probe_map_k, k=kmer_probe_map_k,
                                     with_draws=True)
    uniq, _owner = _unique_last(probe_strs)
    uidx = {p: i for i, p in enumerate(uniq)}
    ent_rank = kmer_entry_ranks(probe_strs, entries, draws, k, hash_fn)
    # equal input probes get equal votes and each counts in the reference's sums
    mult = [0] * len(uniq)
    for p in probe_strs:
        mult[uidx[p]] += 1
    cum = [[0, 0] for _ in uniq]
    for sequence in sequences:
        cov = scan_sequence(sequence, uniq, entries, k, mismatches, lcf_thres,
                            island, merge=True)
        first = scan_first_seen(sequence, uniq, entries, k, mismatches,
                                lcf_thres, island, ent_rank)
        assert set(cov) == set(first)
        # the result dict lists probes in the order they were first met
        order = sorted(cov, key=lambda p: (first[p][0], ent_rank[first[p][1]]))
        intervals = [(r, p) for p in order for r in cov[p]]
        chosen = set(schedule(intervals))
        if detail is not None:
            detail.append(dict(order=order, chosen=sorted(chosen)))
        votes = {p: ((1, 0) if p in chosen else (0, 1)) for p in cov}
        plain = sum(mult[p] * (max(cum[p][0] + a, cum[p][1] + b) - max(cum[p]))
                    for p, (a, b) in votes.items())
        flipped = sum(mult[p] * (max(cum[p][0] + b, cum[p][1] + a) - max(cum[p]))
                      for p, (a, b) in votes.items())
        for p, (a, b) in votes.items():
            if flipped > plain:
                a, b = b, a
            cum[p][0] += a
            cum[p][1] += b
    return [tuple(cum[uidx[p]]) for p in probe_strs]


def adapter_filter(probe_strs, sequences, adapter_a, adapter_b, mismatches,
                   lcf_thres, island=0, kmer_probe_map_k=20, hash_fn=None):
    """AdapterFilter._filter (:363-392): A adapters iff A votes > B votes."""
    votes = adapter_votes(probe_strs, sequences, mismatches, lcf_thres, island,
                          kmer_probe_map_k, hash_fn)
    out = []
    for p, (a, b) in zip(probe_strs, votes):
        five, three = adapter_a if a > b else adapter_b
        out.append(five + p + three)
    return out
