"""FASTA input/output on the host (mirrors catch/utils/seq_io.py:104-252).

These rules define the alphabet the kernels see: read_fasta upper-cases,
maps IUPAC degenerate codes to N and drops '-' (:130, :149-155);
iterate_fasta (used for avoided genomes) maps degenerate codes to N but does
NOT upper-case or drop gaps (:196, :221-224)."""
import gzip
import hashlib
import re
from collections import OrderedDict

from catch_amd import genome

_DEGENERATE = re.compile("[YRWSMKBDHV]")
# the same substitutions as one C-level pass per line (str.translate)
_DEG_TO_N = str.maketrans("YRWSMKBDHV", "N" * 10)
_DEG_TO_N_NO_GAPS = str.maketrans("YRWSMKBDHV", "N" * 10, "-")


def _open(fn):
    return gzip.open(fn, "rt") if fn.endswith(".gz") else open(fn, "r")


def read_fasta(fn, replace_degenerate=True, skip_gaps=True,
               make_uppercase=True):
    """name -> sequence, in file order (catch/utils/seq_io.py:104-175)."""
    m = OrderedDict()
    with _open(fn) as f:
        curr = ""
        for line in f:
            line = line.rstrip()
            if len(line) == 0:
                curr = ""
                continue
            if curr == "":
                assert line.startswith(">")
            if line.startswith(">"):
                curr = line[1:]
                m[curr] = []
            else:
                if make_uppercase:
                    line = line.upper()
                if replace_degenerate and skip_gaps:
                    line = line.translate(_DEG_TO_N_NO_GAPS)
                elif replace_degenerate:
                    line = line.translate(_DEG_TO_N)
                elif skip_gaps:
                    line = line.replace("-", "")
                m[curr].append(line)
    return OrderedDict((k, "".join(v)) for k, v in m.items())


def read_genomes_from_fasta(fn):
    """One Genome per FASTA record (catch/utils/seq_io.py:85-101)."""
    return [genome.Genome.from_one_seq(s) for s in read_fasta(fn).values()]


def iterate_fasta(fn, replace_degenerate=True):
    """Yield each sequence (catch/utils/seq_io.py:178-233)."""
    with _open(fn) as f:
        parts = []
        for line in f:
            line = line.rstrip()
            if len(line) == 0:
                continue
            if line.startswith(">"):
                if parts:
                    yield "".join(parts)
                parts = []
            else:
                if replace_degenerate:
                    line = line.translate(_DEG_TO_N)
                if line:
                    parts.append(line)
        if parts:
            yield "".join(parts)


def write_probe_fasta(probes, out_fn):
    """catch/utils/seq_io.py:235-252 (header = probe.header or the last 10
    hex digits of sha224(sequence), catch/probe.py:303-322)."""
    with open(out_fn, "w") as f:
        for p in probes:
            header = p.header
            if header is None:
                header = hashlib.sha224(p.seq_str.encode()).hexdigest()[-10:]
            f.write(">" + header + "\n")
            f.write(p.seq_str + "\n")
