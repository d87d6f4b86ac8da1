"""Probe value type and the host side of the seed ("anchor") table.

Mirrors the pieces of catch/probe.py that the filters touch: `Probe`
(:38-353: `.seq`, `.seq_str`, hash/eq by sequence, `mismatches`), the anchor
selection of construct_kmer_probe_map_to_find_probe_covers (:507-577, with the
pigeonhole rule :414-504 and the random rule :356-405, which must consume
np.random exactly like the reference for results to be reproducible), and the
module-level pool knob `set_max_num_processes_for_probe_finding_pools`
(:766-779; a no-op here: the scan runs on the GPU).
"""
import numpy as np


class Probe:
    """Immutable probe sequence; equality and hash are by sequence string."""

    __slots__ = ("seq_str", "_seq", "is_flanking_n_string", "header")

    def __init__(self, seq):
        if isinstance(seq, str):
            self.seq_str = seq
            self._seq = None
        else:
            self._seq = seq
            self.seq_str = "".join(seq)
        self.is_flanking_n_string = False
        self.header = None

    @property
    def seq(self):
        if self._seq is None:
            self._seq = np.fromiter(self.seq_str, dtype="U1",
                                    count=len(self.seq_str))
        return self._seq

    @staticmethod
    def from_str(seq_str):
        return Probe(seq_str)

    def mismatches(self, other):
        """catch/probe.py:55-88 (offset 0)."""
        if len(self.seq_str) != len(other.seq_str):
            raise ValueError("Sequences must be of same length")
        a = np.frombuffer(self.seq_str.encode("latin-1"), dtype=np.uint8)
        b = np.frombuffer(other.seq_str.encode("latin-1"), dtype=np.uint8)
        return int(np.count_nonzero(a != b))

    def reverse_complement(self):
        rc_map = {"A": "T", "T": "A", "C": "G", "G": "C"}
        return Probe("".join(rc_map.get(b, b) for b in self.seq_str[::-1]))

    def with_prepended_str(self, s):
        """catch/probe.py:135-146."""
        return Probe(s + self.seq_str)

    def with_appended_str(self, s):
        """catch/probe.py:148-159."""
        return Probe(self.seq_str + s)

    def identifier(self, length=10):
        """Last `length` hex digits of the SHA-224 of the sequence
        (catch/probe.py:301-321)."""
        import hashlib
        return hashlib.sha224(self.seq_str.encode()).hexdigest()[-length:]

    def __hash__(self):
        return hash(self.seq_str)

    def __eq__(self, other):
        return isinstance(other, Probe) and self.seq_str == other.seq_str

    def __len__(self):
        return len(self.seq_str)

    def __getitem__(self, i):
        return self.seq_str[i]

    def __str__(self):
        return self.seq_str

    def __repr__(self):
        return self.seq_str


def set_max_num_processes_for_probe_finding_pools(max_num_processes=8):
    """catch/probe.py:766-779.  Accepted for CLI compatibility; unused."""
    global _pfp_max_num_processes
    _pfp_max_num_processes = max_num_processes


set_max_num_processes_for_probe_finding_pools()


def pigeonhole_kmer_length(probe_length, mismatches):
    """k of _construct_pigeonholed_kmer_probe_map (catch/probe.py:473-491)."""
    if mismatches == 0:
        return probe_length
    k = int(probe_length / mismatches)
    if k == float(probe_length) / mismatches:
        k -= 1
    while probe_length % k != 0:
        k -= 1
    return k


def anchor_table(probe_strs, mismatches, lcf_thres, min_k=20, k=20,
                 num_kmers_per_probe=20, assume_unique=False,
                 with_draws=False):
    """Anchors of construct_kmer_probe_map_to_find_probe_covers
    (catch/probe.py:507-577) for `probe_strs` (duplicates allowed).

    Returns (k, uniq, owner, ent_probe, ent_pos):
      uniq       unique probe strings, first-seen order
      owner      for each unique string the LAST input index holding it (the
                 set id that receives its coverage, catch/filter/
                 set_cover_filter.py:408-412)
      ent_probe, ent_pos   int32 arrays: unique (unique-probe, position)
                 anchors, every anchor k long.
    Random mode (mismatches/lcf_thres None, differing lengths,
    lcf_thres < probe length, or pigeonhole k < min_k) draws
    np.random.choice(n_kmers, size=20, replace=True) once per input probe in
    input order, like catch/probe.py:391-401.
    with_draws: a sixth value, the (input index, position) pairs in the order
    the reference adds them to its k-mer map (one per drawn / pigeonholed
    k-mer, repeats included) -- what the adapter filter needs to reproduce the
    listing order of a k-mer's entries.
    """
    if assume_unique:
        # the caller de-duplicated already: every string owns itself
        uniq = list(probe_strs)
        owner = np.arange(len(uniq), dtype=np.int32)
        first = None
    else:
        first = {}
        last = {}
        for i, p in enumerate(probe_strs):
            if p not in first:
                first[p] = len(first)
            last[p] = i
        uniq = list(first.keys())
        owner = np.fromiter((last[p] for p in uniq), dtype=np.int32,
                            count=len(uniq))
    if not uniq:
        empty = (None, uniq, owner, np.zeros(0, np.int32), np.zeros(0, np.int32))
        return empty + ([],) if with_draws else empty
    draws = []
    L = len(probe_strs[0])
    differ = len(set(map(len, probe_strs))) > 1
    use_random = (mismatches is None or lcf_thres is None or differ
                  or lcf_thres < L)
    kk = None
    if not use_random:
        kk = pigeonhole_kmer_length(L, mismatches)
        if kk < min_k:
            use_random = True
    if use_random:
        kk = k
        # np.random.choice(n, size=20, replace=True) per probe IS randint(0, n,
        # size=20) (legacy RandomState.choice), and legacy randint fills its
        # output element by element from the 32-bit stream, so one randint call
        # over a run of probes with equal k-mer counts draws the same numbers
        # and leaves the generator in the same state as the reference's
        # per-probe calls (tests/test_host_logic.py checks this against the
        # per-probe form)
        nprobes = len(probe_strs)
        lens = np.fromiter(map(len, probe_strs), dtype=np.int64, count=nprobes)
        if (lens < kk).any():
            raise ValueError("k is larger than the length of a probe")
        n_kmers = lens - kk + 1
        pos = np.empty((nprobes, num_kmers_per_probe), dtype=np.int64)
        cuts = np.flatnonzero(np.diff(n_kmers)) + 1
        for a, b in zip([0] + cuts.tolist(), cuts.tolist() + [nprobes]):
            pos[a:b] = np.random.randint(0, int(n_kmers[a]),
                                         size=(b - a, num_kmers_per_probe))
        if first is not None:
            pi = np.fromiter((first[p] for p in probe_strs), dtype=np.int64,
                             count=nprobes)
        else:
            pi = np.arange(nprobes, dtype=np.int64)
        ent_probe, ent_pos = _entries_from_draws(pi, pos)
        if with_draws:
            draws = list(zip(np.repeat(np.arange(nprobes), num_kmers_per_probe).tolist(),
                             pos.ravel().tolist()))
    else:
        per = L // kk
        ent_probe = np.repeat(np.arange(len(uniq), dtype=np.int32), per)
        ent_pos = np.tile(np.arange(0, L, kk, dtype=np.int32), len(uniq))
        if with_draws:
            draws = [(idx, pos) for idx in range(len(probe_strs))
                     for pos in range(0, L, kk)]
    if with_draws:
        return kk, uniq, owner, ent_probe, ent_pos, draws
    return kk, uniq, owner, ent_probe, ent_pos


def anchors_use_random(probe_strs, mismatches, lcf_thres, min_k=20):
    """Whether anchor_table would draw from np.random for these probes (the
    rule of catch/probe.py:507-577): callers that skip or reorder groups must
    then still build the tables in input order, or the stream -- and with it
    the selected probes -- would differ from the reference's."""
    if not probe_strs:
        return False
    L = len(probe_strs[0])
    if (mismatches is None or lcf_thres is None or lcf_thres < L
            or len(set(map(len, probe_strs))) > 1):
        return True
    return pigeonhole_kmer_length(L, mismatches) < min_k


def _entries_from_draws(pi, pos):
    """Sorted unique (probe, position) anchors from the drawn positions
    pos[probe_row][j]; pi[probe_row] = unique-probe index of each row."""
    if pi.size and (pi.size == 1 or bool((np.diff(pi) > 0).all())):
        # one row per probe, rows in probe order: sort inside the rows and drop
        # repeated draws -- no global sort of nprobes * 20 keys
        srt = np.sort(pos, axis=1)
        keep = np.ones(srt.shape, dtype=bool)
        keep[:, 1:] = srt[:, 1:] != srt[:, :-1]
        ent_probe = np.repeat(pi, keep.sum(axis=1)).astype(np.int32)
        return ent_probe, srt[keep].astype(np.int32)
    keys = np.unique(((pi[:, None] << 32) | pos).ravel())
    return (keys >> 32).astype(np.int32), (keys & 0xffffffff).astype(np.int32)


def anchor_entries_equal_length(nprobes, probe_length, mismatches, lcf_thres,
                                min_k=20, k=20, num_kmers_per_probe=20):
    """anchor_table for `nprobes` DISTINCT probes of one length without looking
    at their strings (the device front end never materialises them): returns
    (k, ent_probe, ent_pos) with ent_* = None for the pigeonhole table
    {0, k, 2k, ..} of every probe.  Same rule and the same np.random draws as
    anchor_table(strs, ..., assume_unique=True)."""
    L = probe_length
    use_random = (mismatches is None or lcf_thres is None or lcf_thres < L)
    if not use_random:
        kk = pigeonhole_kmer_length(L, mismatches)
        if kk >= min_k:
            return kk, None, None
    if k > L:
        raise ValueError("k is larger than the length of a probe")
    if nprobes == 0:
        return k, np.zeros(0, np.int32), np.zeros(0, np.int32)
    pos = np.random.randint(0, L - k + 1, size=(nprobes, num_kmers_per_probe))
    ent_probe, ent_pos = _entries_from_draws(np.arange(nprobes, dtype=np.int64), pos)
    return k, ent_probe, ent_pos


def anchor_draws_equal_length(nprobes, probe_length, mismatches, lcf_thres,
                              min_k=20, k=20, num_kmers_per_probe=20):
    """anchor_entries_equal_length with the random anchors left as their draws:
    returns (k, draws) with draws = None for the pigeonhole table, else the
    uint8 array [nprobes][num_kmers_per_probe] of the positions np.random drew
    (the same call, hence the same stream of draws); the device sorts and
    de-duplicates them (engine.Candidates.probes_from_draws).  Only for
    probe_length - k + 1 <= 256 (else use anchor_entries_equal_length)."""
    L = probe_length
    use_random = (mismatches is None or lcf_thres is None or lcf_thres < L)
    if not use_random:
        kk = pigeonhole_kmer_length(L, mismatches)
        if kk >= min_k:
            return kk, None
    if k > L:
        raise ValueError("k is larger than the length of a probe")
    assert L - k + 1 <= 256
    if nprobes == 0:
        return k, np.zeros((0, num_kmers_per_probe), np.uint8)
    pos = np.random.randint(0, L - k + 1, size=(nprobes, num_kmers_per_probe))
    return k, pos.astype(np.uint8)
