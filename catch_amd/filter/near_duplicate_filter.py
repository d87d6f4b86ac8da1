"""Near-duplicate filters on the MI355X: drop-in for
catch/filter/near_duplicate_filter.py.

NearDuplicateFilterWithHammingDistance(dist_thres, probe_length) keeps the
reference's constructor (:115), draws the LSH sampling positions from Python's
`random` in exactly the reference's order (catch/utils/lsh.py:284-287 -> :224
-> :28: per table, k calls of random.randint(0, dim-1)), and resolves the
multiplicity-ordered greedy pass (:60-103) on the GPU
(catchhip_ndf_hamming).  Output: kept probes (subset of the input objects) in
priority order; the reference returns them in CPython-set order, which callers
never rely on (catch/filter/probe_designer.py:288,308).

grouped input is handled in-process by catch_amd's BaseFilter (HIP contexts
do not survive the fork of the reference's BaseFilter)
(HIP contexts do not survive the fork of catch/filter/base_filter.py:121-158).
"""
import logging
import math
import os
import operator
import random
from collections import defaultdict

from catch_amd import _lib
from catch_amd import engine
from catch_amd.filter.base_filter import BaseFilter


logger = logging.getLogger(__name__)


class NearDuplicateFilter(BaseFilter):
    def __init__(self, k, reporting_prob=0.80):
        self.k = k
        self.reporting_prob = reporting_prob

    def _order_by_multiplicity(self, input):
        occurrences = defaultdict(int)
        for p in input:
            occurrences[p] += 1
        # stable sort, ties keep first-seen order (:64-66)
        return [p for p, _ in sorted(occurrences.items(),
                                     key=operator.itemgetter(1),
                                     reverse=True)]


def _as_the_reference_returns_them(kept, key=None):
    """The kept probes (inclusion order in, any sequence type) in the order the
    reference returns them: `list(to_include)`, the iteration order of a SET
    of Probe objects hashed by hash(seq_str)
    (catch/filter/near_duplicate_filter.py:76-103, catch/probe.py:324-329).
    The set cover filter numbers its candidates in the order it gets them, so
    this order decides which of two equally good probes is selected.  It is
    reproducible in the reference only under a fixed PYTHONHASHSEED; the order
    computed here is the one of PYTHONHASHSEED=0 on CPython <= 3.10
    (engine.pyset_order_strs emulates the set), whatever this process' own hash
    seed is -- the same stance as for the MinHash family's hash."""
    if len(kept) < 2 or _lib.test_env("CATCHHIP_NDF_INCLUSION_ORDER"):
        return list(kept)
    order = engine.pyset_order_strs([k if key is None else key(k) for k in kept])
    return [kept[i] for i in order.tolist()]


def _order_strs_by_multiplicity(strs):
    """The same order on plain strings (Counter keeps first-seen order, the
    sort is stable)."""
    from collections import Counter
    return [s for s, _ in sorted(Counter(strs).items(),
                                 key=operator.itemgetter(1), reverse=True)]


class NearDuplicateFilterWithHammingDistance(NearDuplicateFilter):
    def __init__(self, dist_thres, probe_length):
        super().__init__(k=20)
        self.dim = probe_length
        self.dist_thres = dist_thres

    def num_tables(self):
        """catch/utils/lsh.py:268-276."""
        P1 = 1.0 - float(self.dist_thres) / float(self.dim)
        if P1 == 1.0:
            return 1
        return int(math.ceil(math.log(1.0 - self.reporting_prob,
                                      1.0 - math.pow(P1, self.k))))

    def _draw_positions(self):
        return [[random.randint(0, self.dim - 1) for _ in range(self.k)]
                for _ in range(self.num_tables())]

    def _filter(self, input):
        input = list(input)
        order = self._order_by_multiplicity(input)
        positions = self._draw_positions()
        if not order:
            return []
        for p in order:
            # lsh.py:30 asserts len(x) == dim; probe.py:79-80 raises
            if len(p.seq_str) != self.dim:
                raise ValueError("Sequences must be of same length")
        ctx = engine.default_context()
        keep = ctx.ndf_hamming([p.seq_str for p in order], self.dim,
                               positions, self.dist_thres)
        return _as_the_reference_returns_them([p for p, kp in zip(order, keep) if kp],
                                              key=lambda p: p.seq_str)

    def _filter_strs(self, strs):
        """_filter on plain probe strings (the front end's string pipeline)."""
        order = _order_strs_by_multiplicity(strs)
        positions = self._draw_positions()
        if not order:
            return []
        if len(set(map(len, order))) != 1 or len(order[0]) != self.dim:
            raise ValueError("Sequences must be of same length")
        keep = engine.default_context().ndf_hamming(order, self.dim, positions,
                                                    self.dist_thres)
        return _as_the_reference_returns_them([s for s, kp in zip(order, keep) if kp])

    def _draw_for_groups(self, ngroups):
        """The draws of _apply_to_grouped_candidates for ngroups groups, made ahead (callers that run several
        chunks' filters side by side draw for every chunk first, in chunk order: the same stream of draws)."""
        return [self._draw_positions() for _ in range(ngroups)]

    def _apply_to_grouped_candidates(self, cands, ngroups, drawn=None):
        """The same on candidates of grouped targets: sampled positions drawn
        per group in group order, as one _filter call per group would."""
        positions = drawn if drawn is not None else self._draw_for_groups(ngroups)
        if cands.n == 0:
            return
        if cands.L != self.dim:
            raise ValueError("Sequences must be of same length")
        cands.ndf_hamming_many(positions, self.dist_thres)

    def _apply_to_candidates(self, cands):
        """The filter on an engine.Candidates object (device front end):
        multiplicity order, filter and compaction all stay on the device."""
        positions = self._draw_positions()
        if cands.n == 0:
            return
        if cands.L != self.dim:
            raise ValueError("Sequences must be of same length")
        cands.ndf_hamming(positions, self.dist_thres)


class NearDuplicateFilterWithMinHash(NearDuplicateFilter):
    """catch/filter/near_duplicate_filter.py:159-190: MinHash family over
    `kmer_size`-mers (N = 1), k = 3 concatenated functions, Jaccard distance of
    the k-mer sets for verification.

    The reference's family hashes k-mers with the interpreter's salted
    hash(str) (lsh.py:97-104 via use_fast_str_hash=True), so two runs of the
    reference agree only under PYTHONHASHSEED=0; the device computes exactly
    that variant (CPython <= 3.10: SipHash-2-4 with an all-zero key), whatever
    this process' own hash seed is.  (a, b) of every hash function are drawn
    from Python's `random` in the reference's order (lsh.py:284-287 -> :224 ->
    :95-96)."""

    P = 2 ** 31 - 1

    def __init__(self, dist_thres, kmer_size=10):
        super().__init__(k=3)
        self.kmer_size = kmer_size
        self.dist_thres = dist_thres
        self._warn_if_not_reproducible()

    @staticmethod
    def _warn_if_not_reproducible():
        """The device computes the str hash of CPython <= 3.10 under
        PYTHONHASHSEED=0 (SipHash-2-4, zero key).  Under any other interpreter
        state the reference itself would bucket differently (and differently
        from run to run with a random seed): the output is still a valid
        near-duplicate filtering, but not bit-identical to a reference run in
        THIS process -- say so instead of silently computing the 3.10 answer."""
        import os
        import sys
        if getattr(NearDuplicateFilterWithMinHash, "_warned", False):
            return
        NearDuplicateFilterWithMinHash._warned = True
        algo = getattr(sys.hash_info, "algorithm", "")
        seed = os.environ.get("PYTHONHASHSEED", "random")
        if algo != "siphash24" or seed != "0":
            logger.warning(
                "MinHash near-duplicate filter: this interpreter hashes str with %s and "
                "PYTHONHASHSEED=%s; the GPU filter uses SipHash-2-4 with a zero key (CPython "
                "<= 3.10 under PYTHONHASHSEED=0), so its buckets match a reference run only "
                "under that setting", algo or "an unknown algorithm", seed)

    def num_tables(self):
        """lsh.py:268-276 with MinHashFamily.P1 = 1 - dist (:160-174)."""
        P1 = 1.0 - self.dist_thres
        if P1 == 1.0:
            return 1
        return int(math.ceil(math.log(1.0 - self.reporting_prob,
                                      1.0 - math.pow(P1, self.k))))

    def _draw_params(self):
        return [[(random.randint(1, self.P), random.randint(0, self.P))
                 for _ in range(self.k)] for _ in range(self.num_tables())]

    def _filter(self, input):
        input = list(input)
        order = self._order_by_multiplicity(input)
        params = self._draw_params()
        if not order:
            return []
        for p in order:
            # lsh.py:113 asserts kmer_size <= len(s)
            if len(p.seq_str) < self.kmer_size:
                raise AssertionError("k-mer size exceeds a sequence's length")
        ctx = engine.default_context()
        keep = ctx.ndf_minhash([p.seq_str for p in order], self.kmer_size,
                               params, self.dist_thres)
        return _as_the_reference_returns_them([p for p, kp in zip(order, keep) if kp],
                                              key=lambda p: p.seq_str)

    def _filter_strs(self, strs):
        """_filter on plain probe strings (the front end's string pipeline)."""
        order = _order_strs_by_multiplicity(strs)
        params = self._draw_params()
        if not order:
            return []
        if min(map(len, order)) < self.kmer_size:
            raise AssertionError("k-mer size exceeds a sequence's length")
        keep = engine.default_context().ndf_minhash(order, self.kmer_size, params,
                                                    self.dist_thres)
        return _as_the_reference_returns_them([s for s, kp in zip(order, keep) if kp])

    def _draw_for_groups(self, ngroups):
        """See NearDuplicateFilterWithHammingDistance._draw_for_groups.  Many
        groups (the clusters of a clustered design: 2,848 x 25 tables x 3 x 2 =
        427 k `random.randint` calls, 0.3 s of interpreter on the critical path
        of an S5 step) are drawn from a copy of the interpreter's Mersenne
        Twister in NumPy, word for word what `random` would have consumed, and
        `random` is left in the state those calls would have left it in."""
        if ngroups * self.num_tables() * self.k >= 4096:
            try:
                pairs = _randint_pairs_like_random(ngroups * self.num_tables() * self.k, self.P)
            except Exception:      # noqa: BLE001 -- any surprise: the plain calls (the state is untouched until the end)
                pairs = None
            if pairs is not None:
                T, k = self.num_tables(), self.k
                it = iter(pairs)
                return [[[next(it) for _ in range(k)] for _ in range(T)] for _ in range(ngroups)]
        return [self._draw_params() for _ in range(ngroups)]

    def _apply_to_grouped_candidates(self, cands, ngroups, drawn=None):
        """The same on candidates of grouped targets: hash functions drawn per
        group in group order, as one _filter call per group would."""
        params = drawn if drawn is not None else self._draw_for_groups(ngroups)
        if cands.n == 0:
            return
        if cands.L < self.kmer_size:
            raise AssertionError("k-mer size exceeds a sequence's length")
        cands.ndf_minhash_many(self.kmer_size, params, self.dist_thres)

    def _apply_to_candidates(self, cands):
        """The filter on an engine.Candidates object (device front end):
        multiplicity order, filter and compaction all stay on the device."""
        params = self._draw_params()
        if cands.n == 0:
            return
        if cands.L < self.kmer_size:
            raise AssertionError("k-mer size exceeds a sequence's length")
        cands.ndf_minhash(self.kmer_size, params, self.dist_thres)

    def _filter_strs_many(self, groups):
        """_filter_strs for every group (the clusters of a clustered design),
        hash functions drawn per group in group order as one call per group
        would, all groups in one pass over the device."""
        orders, params = [], []
        for strs in groups:
            orders.append(_order_strs_by_multiplicity(strs))
            params.append(self._draw_params())
            if orders[-1] and min(map(len, orders[-1])) < self.kmer_size:
                raise AssertionError("k-mer size exceeds a sequence's length")
        keeps = engine.default_context().ndf_minhash_many(
            orders, self.kmer_size, params, self.dist_thres)
        return [_as_the_reference_returns_them([s for s, kp in zip(order, keep) if kp])
                for order, keep in zip(orders, keeps)]


_bulk_ok = None


def _bulk_draws_match_this_interpreter():
    """the parse below re-implements CPython's private randint -> _randbelow_with_getrandbits word
    consumption.  Checked once per process against the real thing -- a private random.Random() given the caller's
    state draws 64 pairs the slow way; the parse of the same state must give the same pairs and leave the same state
    (the global generator is put back as it was) -- so another interpreter's algorithm cannot silently change the
    hash functions: on any difference every call draws through random.randint itself."""
    global _bulk_ok
    if _bulk_ok is None:
        _bulk_ok = True                # (the parse runs below: do not recurse)
        saved = random.getstate()
        try:
            ref = random.Random()
            ref.setstate(saved)
            P = 2 ** 31 - 1
            want = [(ref.randint(1, P), ref.randint(0, P)) for _ in range(64)]
            got = _randint_pairs_like_random(64, P)
            _bulk_ok = got == want and random.getstate() == ref.getstate()
        except Exception:              # noqa: BLE001 -- anything unexpected: the slow way
            _bulk_ok = False
        finally:
            random.setstate(saved)
    return _bulk_ok


def _randint_pairs_like_random(npairs, P):
    """npairs times (random.randint(1, P), random.randint(0, P)) for P = 2**31 - 1, exactly as the interpreter
    draws them, and `random` advanced as those calls would advance it.  randint(1, P) is 1 + getrandbits(31) -- one
    32-bit output of the generator, shifted (rejected only if it equals P: then this function gives up) --,
    randint(0, P) is getrandbits(32) repeated until it is below 2**31 (CPython's _randbelow_with_getrandbits).  So
    the stream parses as [a][words >= 2**31]*[b] ..., a two-state machine: the state before word i is "next is an
    a" iff the run of words below 2**31 that ends just before i has odd length (even if that run starts the
    stream).  Returns None when `random` is not the Mersenne Twister this relies on."""
    import numpy as np
    st = random.getstate()
    if P != 2 ** 31 - 1 or st[0] != 3 or len(st[1]) != 625:
        return None
    if not _bulk_draws_match_this_interpreter():
        return None
    key, pos = np.array(st[1][:-1], dtype=np.uint32), int(st[1][-1])
    bg = np.random.MT19937()

    def rewind():
        bg.state = {"bit_generator": "MT19937", "state": {"key": key, "pos": pos}}
    R = int(npairs * 3.3) + 64
    while True:
        rewind()
        raw = bg.random_raw(R).astype(np.uint32)
        below = raw < (1 << 31)
        idx = np.arange(R, dtype=np.int64)
        last_not = np.maximum.accumulate(np.where(~below, idx, -1))
        run = np.empty(R + 1, dtype=np.int64)          # words below 2**31 immediately before i
        run[0] = 0
        run[1:] = idx - last_not
        is_a = (run & 1) == 1
        from_start = (np.arange(R + 1, dtype=np.int64) - run) == 0
        is_a = np.where(from_start, ~is_a, is_a)
        is_a[0] = True
        at = np.nonzero(is_a)[0]
        if at.size >= npairs + 1:
            break
        R *= 2
    at = at[:npairs + 1]
    a = raw[at[:-1]] >> 1
    if (a >= P).any():
        return None
    b = raw[at[1:] - 1]
    rewind()
    if int(at[-1]):
        bg.random_raw(int(at[-1]))
    s2 = bg.state["state"]
    out = list(zip((a.astype(np.int64) + 1).tolist(), b.astype(np.int64).tolist()))
    random.setstate((st[0], tuple(int(x) for x in s2["key"]) + (int(s2["pos"]),), st[2]))
    return out
