"""Adds adapters to both ends of each probe (mirrors
catch/filter/adapter_filter.py:120-392).

Which adapter a probe gets is decided by votes: in every target sequence the
probes that hybridize are scheduled as intervals (greedy earliest-end-first,
catch/utils/interval.py:319-358); the scheduled ones vote 'A', the others 'B',
and a sequence's votes are swapped when that raises the total majority
(:299-361).

On the device: the one scan of all sequences at once
(catchhip_cover_scan_first_seen, every sequence its own universe), which also
returns per (probe, sequence) the key of the first accepted seed -- the order
in which the reference's result dict lists probes, and therefore how its
stable sort breaks ties between ranges with equal ends.  At one k-mer position
the reference lists the k-mer's entries by iterating a Python set of
(Probe, position) tuples; the same sets are built here from the caller's probe
objects in the same insertion order, so the interpreter yields the same order
(whatever its string-hash seed is).  The sequential parts run on the device
as well (catchhip_adapter_votes): the rows are sorted by (sequence, end,
first-seen key), one thread per sequence makes the schedule, and one workgroup
walks the sequences in order keeping the running vote totals -- the row table
(hundreds of millions of rows for a large design) never comes to the host.
"""
import logging

import numpy as np

from catch_amd import engine
from catch_amd import probe
from catch_amd.filter.base_filter import BaseFilter

logger = logging.getLogger(__name__)


class AdapterFilter(BaseFilter):
    def __init__(self, adapter_a, adapter_b, mismatches, lcf_thres,
                 island_of_exact_match=0, custom_cover_range_fn=None,
                 kmer_probe_map_k=20):
        if len(adapter_a) != 2 or len(adapter_b) != 2:
            raise ValueError(("adapter_a/adapter_b arguments must be tuples "
                              "of length 2, giving the sequences to add onto "
                              "the 5' and 3' ends"))
        if custom_cover_range_fn is not None:
            raise NotImplementedError(
                "custom hybridization functions cannot run on the GPU path")
        self.adapter_a_5end, self.adapter_a_3end = adapter_a
        self.adapter_b_5end, self.adapter_b_3end = adapter_b
        self.mismatches = mismatches
        self.lcf_thres = lcf_thres
        self.island_of_exact_match = island_of_exact_match
        self.kmer_probe_map_k = kmer_probe_map_k
        # the targets object of the last call: the designer calls the filter
        # once per group of probes with the SAME target genomes
        self._cached_targets = None

    def _targets_for(self, ctx, target_genomes, seqs):
        key = (id(target_genomes), len(seqs), sum(map(len, seqs)))
        if self._cached_targets is not None and self._cached_targets[0] == key:
            return self._cached_targets[1]
        self._drop_cached_targets()
        targets = engine.Targets(ctx, [[s] for s in seqs])
        self._cached_targets = (key, targets, target_genomes)   # keeps the list (and its id) alive
        return targets

    def _drop_cached_targets(self):
        if self._cached_targets is not None:
            self._cached_targets[1].close()
            self._cached_targets = None

    def __del__(self):
        try:
            self._drop_cached_targets()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def _anchor_order(self, probes, strs, uniq, ep, eo, draws, k):
        """For every anchor entry its rank among the entries of its k-mer, in
        the order the reference's map lists them (probe.py:393-401 / :496-503
        build a set of (Probe, pos) per k-mer; SharedKmerProbeMap.construct
        :739-747 iterates it)."""
        uidx = {s: i for i, s in enumerate(uniq)}
        kmer_entries = {}
        for i, pos in draws:
            kmer_entries.setdefault(strs[i][pos:pos + k], set()).add(
                (probes[i], pos))
        rank = {}
        for members in kmer_entries.values():
            for r, (p, pos) in enumerate(members):
                rank[(uidx[p.seq_str], pos)] = r
        return np.fromiter((rank[e] for e in zip(ep.tolist(), eo.tolist())),
                           dtype=np.uint32, count=len(ep))

    def _make_votes_across_target_genomes(self, probes, target_genomes):
        """[(A votes, B votes)] per input probe (:299-361)."""
        strs = [p.seq_str for p in probes]
        if not strs:
            raise ValueError("kmer_probe_map is empty")
        logger.info("Building map from k-mers to probes")
        k, uniq, _owner, ep, eo, draws = probe.anchor_table(
            strs, self.mismatches, self.lcf_thres,
            min_k=self.kmer_probe_map_k, k=self.kmer_probe_map_k,
            with_draws=True)
        order = self._anchor_order(probes, strs, uniq, ep, eo, draws, k)
        uidx = {s: i for i, s in enumerate(uniq)}
        which = np.fromiter((uidx[s] for s in strs), dtype=np.int64,
                            count=len(strs))
        # equal input probes get equal votes and each counts in the sums
        mult = np.bincount(which, minlength=len(uniq)).astype(np.int64)
        seqs = [s for grp in target_genomes for g in grp for s in g.seqs]
        cum_a = np.zeros(len(uniq), dtype=np.int64)
        cum_b = np.zeros(len(uniq), dtype=np.int64)
        if seqs:
            ctx = engine.default_context()
            targets = self._targets_for(ctx, target_genomes, seqs)
            dev = engine.Probes(ctx, uniq,
                                np.arange(len(uniq), dtype=np.int32), ep, eo, k)
            try:
                rows = engine.Rows.scan_first_seen(
                    ctx, dev, targets, self.mismatches, self.lcf_thres,
                    self.island_of_exact_match, 0, engine.SCAN_AUTO, order)
                # scheduling inside every sequence and the running totals over
                # the sequences, on the device (catchhip_adapter_votes)
                cum_a, cum_b = rows.adapter_votes(mult)
                rows.close()
            finally:
                dev.close()
        return list(zip(cum_a[which].tolist(), cum_b[which].tolist()))

    def _filter(self, input, target_genomes):
        input = list(input)
        logger.info("Computing adapter votes across all target genomes")
        votes = self._make_votes_across_target_genomes(input, target_genomes)
        logger.info("Adding adapters to probes based on votes")
        out = []
        for p, (a, b) in zip(input, votes):
            if a > b:
                out.append(p.with_prepended_str(self.adapter_a_5end)
                           .with_appended_str(self.adapter_a_3end))
            else:
                out.append(p.with_prepended_str(self.adapter_b_5end)
                           .with_appended_str(self.adapter_b_3end))
        return out
