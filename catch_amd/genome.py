"""Genome value type at the filter boundary (mirrors catch/genome.py:9-143:
`.seqs` list of sequence strings, `.size()`)."""
from collections import OrderedDict


_DROP_ATCG = str.maketrans("", "", "ATCG")


class Genome:
    __slots__ = ("seqs", "chrs", "_size", "_size_unambig")      # (a clustered design makes one per fragment: 224 k at S5)

    def __init__(self, seqs, chrs=None):
        if len(seqs) > 1 and chrs is None:
            raise ValueError(("When there is more than one sequence, chrs "
                              "should also be specified"))
        self.seqs = seqs
        self.chrs = chrs
        self._size = None
        self._size_unambig = None

    def divided_into_chrs(self):
        return len(self.seqs) > 1

    def size(self, only_unambig=False):
        if only_unambig:
            if self._size_unambig is None:
                # bases that are A, T, C or G (catch/genome.py:52-56), one pass
                self._size_unambig = sum(
                    len(seq) - len(seq.translate(_DROP_ATCG))
                    for seq in self.seqs)
            return self._size_unambig
        if self._size is None:
            self._size = sum(len(seq) for seq in self.seqs)
        return self._size

    def break_into_fragments(self, fragment_length, include_full_end=False):
        """New Genome whose sequences are these cut into pieces of
        fragment_length; with include_full_end a short last piece is replaced
        by the last fragment_length bases (catch/genome.py:64-100)."""
        def pieces(seq):
            for i in range(0, len(seq), fragment_length):
                piece = seq[i:i + fragment_length]
                if include_full_end and len(piece) < fragment_length:
                    piece = seq[max(0, len(seq) - fragment_length):]
                yield piece
        out = OrderedDict()
        if self.chrs is None:
            assert len(self.seqs) == 1
            for idx, piece in enumerate(pieces(self.seqs[0])):
                out[str(idx)] = piece
        else:
            for name, seq in self.chrs.items():
                for idx, piece in enumerate(pieces(seq)):
                    out[name + "-" + str(idx)] = piece
        return Genome.from_chrs(out)

    def __hash__(self):
        return hash(tuple(self.seqs))

    def __eq__(self, other):
        return (isinstance(other, Genome) and self.seqs == other.seqs
                and self.chrs == other.chrs)

    @staticmethod
    def from_chrs(seqs_by_chr):
        seqs = list(seqs_by_chr.values())
        return Genome(seqs, OrderedDict(seqs_by_chr))

    @staticmethod
    def from_one_seq(seq):
        # (no argument checks to make, and the size is known: a clustered design builds one per fragment)
        g = object.__new__(Genome)
        g.seqs = [seq]
        g.chrs = None
        g._size = len(seq)
        g._size_unambig = None
        return g
