"""Candidate probes by sliding window.

Mirrors the interface of catch/filter/candidate_probes.py (:21-182): windows
of probe_length every probe_stride bases, one more flush with the end when the
length is not a multiple of the stride, windows holding a run of at least
min_n_string_length N's dropped, and the windows flanking every such run
added.  Here the windows are computed once, as (start, is_flank) pairs by
interval arithmetic over the N runs (`window_starts`); the string and the
Probe-object forms are thin layers over that list, and
catch_amd/csrc/candidates.hip enumerates the same list on the device.
"""
import re

from catch_amd import probe


def _n_runs(seq, min_n):
    """[a, b) of every maximal run of at least min_n N's."""
    if "N" * min_n not in seq:
        return []
    return [(m.start(), m.end()) for m in re.finditer("N{%d,}" % min_n, seq)]


def window_starts(seq, L, stride, min_n=2):
    """(start, flanks_a_run) of every candidate window of a sequence of at
    least L bases, in the reference's order: stride windows, the end-flush
    window, then for every N run its left and right neighbour."""
    n = len(seq)
    runs = _n_runs(seq, min_n)
    # a window [s, s+L) holds min_n consecutive N's of run [a, b)
    # iff a + min_n - L <= s <= b - min_n
    blocked = [(a + min_n - L, b - min_n) for a, b in runs]

    def clean(s):
        return not any(lo <= s <= hi for lo, hi in blocked)

    starts = list(range(0, n - L + 1, stride))
    if n % stride:
        starts.append(n - L)
    out = [(s, False) for s in starts if clean(s)] if blocked else \
        [(s, False) for s in starts]
    for a, b in runs:
        for s in (a - L, b):
            if 0 <= s <= n - L and clean(s):
                out.append((s, True))
    return out


def _short_sequence(seq, L, min_n, allow_small_seqs):
    """A sequence shorter than the probe length is its own single candidate
    when --small-seq-min allows it (:64-83)."""
    if not allow_small_seqs:
        raise ValueError("sequence of %d bases is shorter than the probe length "
                         "%d (--small-seq-skip leaves such sequences out)"
                         % (len(seq), L))
    if allow_small_seqs > len(seq):
        raise ValueError("sequence of %d bases is below the --small-seq-min "
                         "of %d" % (len(seq), allow_small_seqs))
    if _n_runs(seq, min_n):
        raise Exception("the one candidate of a short sequence contains a run "
                        "of N's")
    return seq


def _check_list(seqs):
    """Same exception types as the reference for the same mistakes (:150-160)."""
    if type(seqs) is not list and not isinstance(seqs, list):
        raise TypeError("expected a list of sequence strings, got %s" % type(seqs).__name__)
    if not seqs:
        raise ValueError("no sequences to make candidate probes from")
    bad = next((x for x in seqs if not isinstance(x, str)), None)
    if bad is not None:
        raise TypeError("every sequence must be a str, found %s" % type(bad).__name__)


def candidate_strings_from_sequences(seqs, probe_length, probe_stride, min_n_string_length=2,
                                     allow_small_seqs=None, seq_length_to_skip=None):
    """Candidate windows of all sequences as plain strings, in the order the
    reference generates its Probe objects."""
    _check_list(seqs)
    L, stride = probe_length, probe_stride
    out = []
    for seq in seqs:
        n = len(seq)
        if seq_length_to_skip is not None and n <= seq_length_to_skip:
            continue
        if n < L:
            out.append(_short_sequence(seq, L, min_n_string_length, allow_small_seqs))
        elif "N" * min_n_string_length not in seq:
            out += [seq[i:i + L] for i in range(0, n - L + 1, stride)]
            if n % stride:
                out.append(seq[n - L:])
        else:
            out += [seq[s:s + L] for s, _ in
                    window_starts(seq, L, stride, min_n_string_length)]
    return out


def make_candidate_probes_from_sequence(seq, probe_length, probe_stride, min_n_string_length=2,
                                        allow_small_seqs=None):
    """Probe objects of one sequence (is_flanking_n_string set on the windows
    next to an N run)."""
    if not isinstance(seq, str):
        seq = "".join(seq)
    if len(seq) < probe_length:
        return [probe.Probe.from_str(_short_sequence(
            seq, probe_length, min_n_string_length, allow_small_seqs))]
    out = []
    for s, flank in window_starts(seq, probe_length, probe_stride,
                                  min_n_string_length):
        p = probe.Probe.from_str(seq[s:s + probe_length])
        p.is_flanking_n_string = flank
        out.append(p)
    return out


def make_candidate_probes_from_sequences(seqs, probe_length, probe_stride, min_n_string_length=2,
                                         allow_small_seqs=None, seq_length_to_skip=None):
    _check_list(seqs)
    out = []
    for one in seqs:
        if seq_length_to_skip is not None and len(one) <= seq_length_to_skip:
            continue
        seq = one
        out += make_candidate_probes_from_sequence(
            seq, probe_length, probe_stride, min_n_string_length,
            allow_small_seqs)
    return out
