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
