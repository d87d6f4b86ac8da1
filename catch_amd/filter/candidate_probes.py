"""Candidate probes by sliding window (catch/filter/candidate_probes.py
:21-182): windows of probe_length every probe_stride bases, a last window
flush with the end when the length is not a multiple of the stride, windows
containing a run of >= min_n_string_length N dropped, and windows flanking
every such run added."""
import re

from catch_amd import probe


def make_candidate_probes_from_sequence(seq, probe_length, probe_stride,
                                        min_n_string_length=2,
                                        allow_small_seqs=None):
    n_string_query = re.compile("(N{" + str(min_n_string_length) + ",})")
    if len(seq) < probe_length:
        if allow_small_seqs:
            if len(seq) < allow_small_seqs:
                raise ValueError(("Allowing sequences smaller than the probe "
                                  "length (" + str(probe_length) + "), but "
                                  "input sequence is smaller than minimum "
                                  "allowed length"))
            if n_string_query.search(seq):
                raise Exception(("Only possible probe from input "
                                 "sequence has too long a stretch of N's"))
            return [probe.Probe.from_str(seq)]
        raise ValueError(("An input sequence is smaller than the probe "
                          "length (" + str(probe_length) + "); try "
                          "setting --small-seq-skip"))
    if not isinstance(seq, str):
        seq = "".join(seq)

    probes = []

    def add(start, end, flank=False):
        sub = seq[start:end]
        if not n_string_query.search(sub):
            p = probe.Probe.from_str(sub)
            p.is_flanking_n_string = flank
            probes.append(p)

    for start in range(0, len(seq), probe_stride):
        if start + probe_length > len(seq):
            break
        add(start, start + probe_length)
    if len(seq) % probe_stride != 0:
        add(len(seq) - probe_length, len(seq))
    for match in n_string_query.finditer(seq):
        if match.start() - probe_length >= 0:
            add(match.start() - probe_length, match.start(), True)
        if match.end() + probe_length <= len(seq):
            add(match.end(), match.end() + probe_length, True)
    return probes


def candidate_strings_from_sequences(seqs, probe_length, probe_stride,
                                     min_n_string_length=2,
                                     allow_small_seqs=None,
                                     seq_length_to_skip=None):
    """The sequences of make_candidate_probes_from_sequences(...) as plain
    strings, same content and order, without building Probe objects.  A
    sequence without a run of min_n_string_length N's (the usual case) is
    sliced directly; one with such runs goes through the per-window code."""
    if not isinstance(seqs, list):
        raise TypeError("seqs must be a list of sequences")
    if len(seqs) == 0:
        raise ValueError("seqs must have at least one sequence")
    n_run = "N" * min_n_string_length
    n_query = re.compile("(N{" + str(min_n_string_length) + ",})")
    L, stride = probe_length, probe_stride
    out = []
    for seq in seqs:
        if not isinstance(seq, str):
            raise TypeError("seqs must be a list of Python strings")
        if seq_length_to_skip is not None and len(seq) <= seq_length_to_skip:
            continue
        n = len(seq)
        if n >= L and n_run not in seq:
            out += [seq[i:i + L] for i in range(0, n - L + 1, stride)]
            if n % stride != 0:
                out.append(seq[n - L:])
        elif n >= L:
            # runs of >= min_n N's: a window [s, s+L) holds min_n consecutive
            # N's of the run [a, b) iff a + min_n - L <= s <= b - min_n
            runs = [(m.start(), m.end()) for m in n_query.finditer(seq)]
            bad = [(a + min_n_string_length - L, b - min_n_string_length)
                   for a, b in runs]

            def ok(s0):
                for lo, hi in bad:
                    if lo <= s0 <= hi:
                        return False
                return True
            out += [seq[i:i + L] for i in range(0, n - L + 1, stride) if ok(i)]
            if n % stride != 0 and ok(n - L):
                out.append(seq[n - L:])
            for a, b in runs:     # windows flanking every run
                if a - L >= 0 and ok(a - L):
                    out.append(seq[a - L:a])
                if b + L <= n and ok(b):
                    out.append(seq[b:b + L])
        else:
            out += [p.seq_str for p in make_candidate_probes_from_sequence(
                seq, probe_length=L, probe_stride=stride,
                min_n_string_length=min_n_string_length,
                allow_small_seqs=allow_small_seqs)]
    return out


def make_candidate_probes_from_sequences(seqs, probe_length, probe_stride,
                                         min_n_string_length=2,
                                         allow_small_seqs=None,
                                         seq_length_to_skip=None):
    if not isinstance(seqs, list):
        raise TypeError("seqs must be a list of sequences")
    if len(seqs) == 0:
        raise ValueError("seqs must have at least one sequence")
    for seq in seqs:
        if not isinstance(seq, str):
            raise TypeError("seqs must be a list of Python strings")
    probes = []
    for seq in seqs:
        if seq_length_to_skip is not None and len(seq) <= seq_length_to_skip:
            continue
        probes += make_candidate_probes_from_sequence(
            seq, probe_length=probe_length, probe_stride=probe_stride,
            min_n_string_length=min_n_string_length,
            allow_small_seqs=allow_small_seqs)
    return probes
