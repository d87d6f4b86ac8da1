"""MinHash family for sequence signatures on the MI355X (mirrors
lsh.MinHashFamily of catch/utils/lsh.py:48-215 with the deterministic md5
inner hash, the form the clustering pre-step uses).

`make_h()` draws (a, b) exactly as the reference does (random.randint(1, p),
random.randint(0, p), :95-96) and returns a function whose signatures are
computed by catchhip_sigs_create; `signatures()` is the batched form the
clustering code calls (one upload, one kernel sequence, signatures stay on the
device for the distance kernels).
"""
import logging
import random

from catch_amd import engine

logger = logging.getLogger(__name__)

P = 2 ** 31 - 1


class MinHashFamily:
    def __init__(self, kmer_size, N=1, use_fast_str_hash=False):
        if use_fast_str_hash:
            # hash(str) varies across processes; the clustering path never uses it
            raise NotImplementedError(
                "use_fast_str_hash is only available inside "
                "NearDuplicateFilterWithMinHash")
        self.kmer_size = kmer_size
        self.N = N
        self.use_fast_str_hash = False

    def _draw(self):
        a = random.randint(1, P)
        b = random.randint(0, P)
        return a, b

    def _warn(self, s):
        if self.kmer_size >= len(s) / 2:
            logger.warning(("The k-mer size %d is large (> (1/2)x) compared "
                            "to the size of a sequence to hash (%d), which "
                            "might make it difficult for MinHash to find "
                            "similar sequence"), self.kmer_size, len(s))
        num_kmers = len(s) - self.kmer_size + 1
        if num_kmers < self.N:
            logger.warning(("The number of k-mers (%d) in a given sequence is "
                            "too small to produce a signature of size %d; the "
                            "MinHash family might provide unreliable distances "
                            "against the sequence. This might be fine, or "
                            "specify --small-seq-skip to skip the sequence."),
                           num_kmers, self.N)

    def signatures(self, seqs, ctx=None, ab=None):
        """engine.Signatures of `seqs` under one freshly drawn hash function
        (or under ab = (a, b))."""
        a, b = ab if ab is not None else self._draw()
        if isinstance(seqs, engine.FragmentTable):
            # (views of the parents' storage: the same checks on the lengths, the same warnings once per kind)
            if len(seqs):
                assert self.kmer_size <= int(seqs.length.min())
                self._warn(range(int(seqs.length.min())))       # (the shortest one: _warn only looks at the length)
        else:
            for s in seqs:
                assert self.kmer_size <= len(s)
                self._warn(s)
        ctx = ctx or engine.default_context()
        return engine.Signatures(ctx, seqs, self.kmer_size, self.N, a, b)

    def make_h(self):
        ab = self._draw()

        def h(s):
            sigs = self.signatures([s], ab=ab)
            try:
                return tuple(int(x) for x in sigs.fetch()[0])
            finally:
                sigs.close()
        return h

    def P1(self, dist):
        return 1.0 - dist

    def estimate_jaccard_dist(self, hA, hB):
        """Distance of two signatures (:170-215); the clustering path computes
        this for whole rows / all pairs on the device (Signatures.common_row,
        Signatures.condensed) -- this two-signature form exists for callers
        that hold signatures as tuples."""
        i = j = common = steps = 0
        while i < len(hA) and j < len(hB) and steps < self.N:
            if hA[i] < hB[j]:
                i += 1
            elif hA[i] > hB[j]:
                j += 1
            else:
                common += 1
                i += 1
                j += 1
            steps += 1
        return 1.0 - float(common) / steps
