"""Coverage analysis of a probe set on the MI355X: drop-in for
catch/coverage_analysis.py (Analyzer, :66-600).

The one expensive step -- finding every range of every target genome (and of
its reverse complement) that some probe covers,
`probe.find_probe_covers_in_sequence(sequence, merge_overlapping=False)` per
sequence (:183-280) -- is one catchhip_cover_ranges call per strand over all
genomes at once (every sequence its own universe); the statistics the
reference derives from those ranges (:282-335) are reduced on the device too (catchhip_rows_stats: bases
covered, summed range lengths, sequences per probe), so a report over thousands
of genomes never brings the row table to the host; `target_covers` and
`sliding_coverage` (:337-411) are fetched / computed on first use.
Same constructor, attributes (`target_covers`, `bp_covered`,
`average_coverage`, `sliding_coverage`, `probe_map_counts`) and writers.

Not supported (raises NotImplementedError): `custom_cover_range_fn`.
"""
from collections import Counter
import logging

import numpy as np

from catch_amd import engine
from catch_amd import probe

logger = logging.getLogger(__name__)

_RC = str.maketrans("ACGT", "TGCA")


class Analyzer:
    def __init__(self, probes, mismatches, lcf_thres, target_genomes,
                 target_genomes_names=None, island_of_exact_match=0,
                 custom_cover_range_fn=None, cover_extension=0,
                 kmer_probe_map_k=10, rc_too=True):
        if custom_cover_range_fn is not None:
            raise NotImplementedError(
                "custom hybridization functions cannot run on the GPU path")
        self.probes = probes
        self.target_genomes = target_genomes
        if target_genomes_names:
            if len(target_genomes_names) != len(target_genomes):
                raise ValueError(("Number of target genome names must be same "
                                  "as the number of target genomes"))
            self.target_genomes_names = target_genomes_names
        else:
            self.target_genomes_names = ["Group %d" % i
                                         for i in range(len(target_genomes))]
        self.mismatches = mismatches
        self.lcf_thres = lcf_thres
        self.island_of_exact_match = island_of_exact_match
        self.cover_extension = cover_extension
        self.kmer_probe_map_k = kmer_probe_map_k
        self.rc_too = rc_too
        self._covers = None
        self._sliding = None
        self._window = (50, 25)
        self._scan_inputs = None

    def _iter_target_genomes(self):
        for i, genomes_from_group in enumerate(self.target_genomes):
            for j, gnm in enumerate(genomes_from_group):
                yield i, j, gnm, False
                if self.rc_too:
                    yield i, j, gnm, True

    # ------------------------------------------------------------------
    def _strand_targets(self, ctx, flat, rc):
        """All sequences of all genomes as one targets object, every sequence
        its own universe (ranges stay per sequence); plus, per sequence, the
        genome it belongs to and its offset inside the genome."""
        seqs, owner_of_seq, offset_of_seq = [], [], []
        for g, (_i, _j, gnm) in enumerate(flat):
            so_far = 0
            for s in gnm.seqs:
                seqs.append([s[::-1].translate(_RC) if rc else s])
                owner_of_seq.append(g)
                offset_of_seq.append(so_far)
                so_far += len(s)
        return (engine.Targets(ctx, seqs), np.asarray(owner_of_seq, dtype=np.int64),
                np.asarray(offset_of_seq, dtype=np.int64))

    def _scan(self, ctx, probes_dev, targets):
        return engine.Rows.scan(ctx, probes_dev, targets, self.mismatches,
                                self.lcf_thres, self.island_of_exact_match,
                                self.cover_extension, engine.SCAN_AUTO, merge=False)

    def _flat_genomes(self):
        return [(i, j, gnm) for i, grp in enumerate(self.target_genomes)
                for j, gnm in enumerate(grp)]

    def _find_covers_in_target_genomes(self):
        """One scan per strand over all genomes (every distinct range of every
        probe, nothing merged: find_probe_covers_in_sequence(...,
        merge_overlapping=False), :183-280).  The row table stays on the device;
        what the report needs is reduced there (catchhip_rows_stats):
          self.bp_covered[i][j][rc]   bases covered by at least one probe (:282-302)
          self._total_covered[i][j][rc]  sum of the ranges' lengths (:318-320)
          self.probe_map_counts[p]    sequences probe p maps to, forward strand (:255-258)
        self.target_covers / self.sliding_coverage are fetched / computed on
        first use (they need every range on the host)."""
        logger.info("Finding probe covers across target genomes")
        self.probe_map_counts = Counter()
        self.bp_covered, self._total_covered = {}, {}
        for i, j, _gnm, rc in self._iter_target_genomes():
            for d in (self.bp_covered, self._total_covered):
                d.setdefault(i, {}).setdefault(j, {False: None, True: None})[rc] = 0
        self._covers = None
        strs = [p.seq_str for p in self.probes]
        flat = self._flat_genomes()
        self._scan_inputs = None
        if not strs or not flat:
            return
        ctx = engine.default_context()
        self._scan_inputs = probe.anchor_table(
            strs, self.mismatches, self.lcf_thres,
            min_k=self.kmer_probe_map_k, k=self.kmer_probe_map_k)
        k, uniq, owner, ep, eo = self._scan_inputs
        probes_dev = engine.Probes(ctx, uniq, owner, ep, eo, k)
        try:
            for rc in ((False, True) if self.rc_too else (False,)):
                targets, owner_of_seq, _off = self._strand_targets(ctx, flat, rc)
                try:
                    rows = self._scan(ctx, probes_dev, targets)
                    total_len, union_len, per_probe = rows.stats(
                        targets.ngenomes, len(strs) if not rc else 0)
                    rows.close()
                finally:
                    targets.close()
                tot = np.bincount(owner_of_seq, weights=total_len.astype(np.float64),
                                  minlength=len(flat)).astype(np.int64)
                uni = np.bincount(owner_of_seq, weights=union_len.astype(np.float64),
                                  minlength=len(flat)).astype(np.int64)
                for g, (i, j, _gnm) in enumerate(flat):
                    self.bp_covered[i][j][rc] = int(uni[g])
                    self._total_covered[i][j][rc] = int(tot[g])
                if not rc:
                    for pi in np.nonzero(per_probe)[0]:
                        self.probe_map_counts[self.probes[pi]] += int(per_probe[pi])
        finally:
            probes_dev.close()

    @property
    def target_covers(self):
        """target_covers[i][j][rc] = sorted list of (start, end) in genome
        coordinates (chromosomes offset by the lengths before them), one entry
        per distinct range of every probe.  Fetched from the device on first
        use (a second scan: the report itself does not need them)."""
        if self._covers is None:
            self._covers = self._fetch_covers()
        return self._covers

    def _fetch_covers(self):
        covers = {}
        for i, j, _gnm, rc in self._iter_target_genomes():
            covers.setdefault(i, {}).setdefault(j, {False: None, True: None})[rc] = []
        flat = self._flat_genomes()
        if self._scan_inputs is None:
            return covers
        ctx = engine.default_context()
        k, uniq, owner, ep, eo = self._scan_inputs
        probes_dev = engine.Probes(ctx, uniq, owner, ep, eo, k)
        try:
            for rc in ((False, True) if self.rc_too else (False,)):
                targets, owner_of_seq, offset_of_seq = self._strand_targets(ctx, flat, rc)
                try:
                    rows = self._scan(ctx, probes_dev, targets)
                    _sid, univ, st, en = rows.fetch()
                    rows.close()
                finally:
                    targets.close()
                gidx = owner_of_seq[univ] if univ.size else univ.astype(np.int64)
                a = st + (offset_of_seq[univ] if univ.size else 0)
                b = en + (offset_of_seq[univ] if univ.size else 0)
                order = np.lexsort((b, a, gidx))
                gidx, a, b = gidx[order], a[order], b[order]
                bounds = np.searchsorted(gidx, np.arange(len(flat) + 1))
                for g, (i, j, _gnm) in enumerate(flat):
                    lo, hi = bounds[g], bounds[g + 1]
                    covers[i][j][rc] = list(zip(a[lo:hi].tolist(), b[lo:hi].tolist()))
        finally:
            probes_dev.close()
        return covers

    def _compute_bp_covered_in_target_genomes(self):
        """Done on the device by _find_covers_in_target_genomes."""

    def _compute_average_coverage_in_target_genomes(self):
        self.average_coverage = {}
        for i, j, gnm, rc in self._iter_target_genomes():
            total_covered = self._total_covered[i][j][rc]
            self.average_coverage.setdefault(i, {}).setdefault(
                j, {False: None, True: None})[rc] = (
                float(total_covered) / gnm.size(False),
                float(total_covered) / gnm.size(True))

    def _compute_sliding_coverage_in_target_genomes(self, window_length,
                                                    window_stride):
        """:337-411: average depth in windows; keys are window middles.
        Needs every range on the host: computed on first use of
        self.sliding_coverage."""
        self._window = (window_length, window_stride)
        self._sliding = None

    @property
    def sliding_coverage(self):
        if self._sliding is None:
            window_length, window_stride = self._window
            self._sliding = {}
            for i, j, gnm, rc in self._iter_target_genomes():
                covers = self.target_covers[i][j][rc]
                n = gnm.size(False)
                diff = np.zeros(n + 1, dtype=np.int64)
                if covers:
                    arr = np.asarray(covers, dtype=np.int64)
                    np.add.at(diff, arr[:, 0], 1)
                    np.add.at(diff, arr[:, 1], -1)
                # the reference stores the depth as uint16
                counts = np.cumsum(diff[:n]).astype(np.uint16)
                out = {}
                for window_start in np.arange(0, n, window_stride):
                    window_end = window_start + window_length
                    if window_end > n:
                        window_end = n
                        window_start = window_end - window_length
                    middle = window_start + (window_length / 2)
                    out[middle] = np.average(counts[window_start:window_end])
                self._sliding.setdefault(i, {}).setdefault(
                    j, {False: None, True: None})[rc] = out
        return self._sliding

    def run(self, window_length=50, window_stride=25):
        self._find_covers_in_target_genomes()
        self._compute_bp_covered_in_target_genomes()
        self._compute_average_coverage_in_target_genomes()
        self._compute_sliding_coverage_in_target_genomes(window_length,
                                                         window_stride)

    # ------------------------------------------------------------------
    def _row_header(self, i, j, rc):
        h = "%s, genome %d" % (self.target_genomes_names[i], j)
        return h + " (rc)" if rc else h

    def write_data_matrix_as_tsv(self, fn):
        data = [["Genome", "Num bases covered", "Frac bases covered",
                 "Frac bases covered over unambig", "Average coverage/depth",
                 "Average coverage/depth over unambig"]]
        for i, j, gnm, rc in self._iter_target_genomes():
            bp_covered = self.bp_covered[i][j][rc]
            avg_all, avg_unambig = self.average_coverage[i][j][rc]
            data.append([self._row_header(i, j, rc), bp_covered,
                         float(bp_covered) / gnm.size(False),
                         float(bp_covered) / gnm.size(True),
                         avg_all, avg_unambig])
        with open(fn, "w") as f:
            for row in data:
                f.write("\t".join(str(entry) for entry in row) + "\n")

    def _make_data_matrix_string(self):
        data = [["Genome", "Num bases covered\n[over unambig]",
                 "Average coverage/depth\n[over unambig]"]]
        for i, j, gnm, rc in self._iter_target_genomes():
            bp_covered = self.bp_covered[i][j][rc]
            frac_all = float(bp_covered) / gnm.size(False)
            frac_unambig = float(bp_covered) / gnm.size(True)
            all_str = "<0.01%" if frac_all < 0.0001 else "{0:.2%}".format(frac_all)
            unambig_str = ("<0.01%" if frac_unambig < 0.0001
                           else "{0:.2%}".format(frac_unambig))
            bp_str = "%d (%s) [%s]" % (bp_covered, all_str, unambig_str)
            avg_all, avg_unambig = self.average_coverage[i][j][rc]
            a = "<0.01" if avg_all < 0.01 else "{0:.2f}".format(avg_all)
            u = "<0.01" if avg_unambig < 0.01 else "{0:.2f}".format(avg_unambig)
            data.append([self._row_header(i, j, rc), bp_str, "%s [%s]" % (a, u)])
        return data

    def print_analysis(self):
        print("NUMBER OF PROBES: %d" % len(self.probes))
        print()
        print(_table(self._make_data_matrix_string(),
                     ["left", "right", "right"]))

    def write_sliding_window_coverage(self, fn):
        with open(fn, "w") as f:
            for i, j, _gnm, rc in self._iter_target_genomes():
                header = self._row_header(i, j, rc)
                cov = self.sliding_coverage[i][j][rc]
                for pos in sorted(cov.keys()):
                    f.write("\t".join(str(x) for x in [header, pos, cov[pos]]) + "\n")

    def write_probe_map_counts(self, fn):
        with open(fn, "w") as f:
            f.write("\t".join(["Probe identifier", "Probe sequence",
                               "Number sequences mapped to"]) + "\n")
            for p, count in self.probe_map_counts.items():
                ident = p.identifier() if hasattr(p, "identifier") else ""
                f.write("\t".join(str(x) for x in [ident, p.seq_str, count]) + "\n")


def _table(data, col_justify):
    """Plain-text table with multi-line cells and an underlined header
    (layout of catch/utils/pretty_print.py:7-90)."""
    cells = [[str(c).rstrip().split("\n") for c in row] for row in data]
    widths = [max(len(line) for row in cells for line in row[j])
              for j in range(len(data[0]))]
    out = []
    for r, row in enumerate(cells):
        height = max(len(c) for c in row)
        for h in range(height):
            parts = []
            for j, c in enumerate(row):
                text = c[h] if h < len(c) else ""
                if col_justify[j] == "right":
                    parts.append(text.rjust(widths[j]))
                elif col_justify[j] == "center":
                    parts.append(text.center(widths[j]))
                else:
                    parts.append(text.ljust(widths[j]))
            out.append(" ".join(parts).rstrip())
        if r == 0:
            out.append(" ".join("-" * w for w in widths))
    return "\n".join(out)
