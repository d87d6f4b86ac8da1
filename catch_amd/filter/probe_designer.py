"""ProbeDesigner for the unclustered case: candidate probes per group, then the
filter list over grouped input (mirrors catch/filter/probe_designer.py
:186-207, :230-271, :273-289; `--cluster-and-design-separately` is out of scope
of this round)."""
import itertools
import logging

from catch_amd import probe
from catch_amd.filter import candidate_probes
from catch_amd.filter.duplicate_filter import DuplicateFilter
from catch_amd.filter.near_duplicate_filter import (
    NearDuplicateFilterWithHammingDistance, NearDuplicateFilterWithMinHash)
from catch_amd.filter.set_cover_filter import SetCoverFilter

logger = logging.getLogger(__name__)


class ProbeDesigner:
    def __init__(self, genomes, filters, probe_length, probe_stride,
                 allow_small_seqs=None, seq_length_to_skip=None,
                 cluster_threshold=None, **_unused):
        if cluster_threshold is not None:
            raise NotImplementedError(
                "--cluster-and-design-separately is not built yet")
        self.genomes = genomes
        self.filters = filters
        self.probe_length = probe_length
        self.probe_stride = probe_stride
        self.allow_small_seqs = allow_small_seqs
        self.seq_length_to_skip = seq_length_to_skip
        self._candidates = None
        self._candidate_strs = None
        self.final_probes = None

    def _pass_through_filters(self, probes, genomes, filters):
        assert len(probes) == len(genomes)
        for f in filters:
            logger.info("Starting filter %s", f.__class__.__name__)
            probes = f.filter(probes, genomes, input_is_grouped=True)
        return probes

    def _design_for_genomes(self, genomes, filters):
        candidates = []
        for genomes_from_group in genomes:
            c = []
            for g in genomes_from_group:
                c += candidate_probes.make_candidate_probes_from_sequences(
                    g.seqs, probe_length=self.probe_length,
                    probe_stride=self.probe_stride,
                    allow_small_seqs=self.allow_small_seqs,
                    seq_length_to_skip=self.seq_length_to_skip)
            if len(c) == 0:
                logger.warning("There are no candidate probes for a grouping "
                               "of genomes")
            candidates.append(c)
        return candidates, self._pass_through_filters(candidates, genomes,
                                                      filters)

    def _design_on_strings(self):
        """[DuplicateFilter | near-duplicate filter, SetCoverFilter] -- the
        filter lists bin/design.py:296-340 builds -- on plain strings:
        candidates are sliced, de-duplicated (dict, or the LSH filter on the
        device) and handed to the set cover filter without a Probe object per
        candidate (a design over 8,000 genomes spent 0.85 of its 1.0 s building
        them); only the selected probes become objects."""
        first, scf = self.filters
        cand = []
        for genomes_from_group in self.genomes:
            c = []
            for g in genomes_from_group:
                c += candidate_probes.candidate_strings_from_sequences(
                    list(g.seqs), probe_length=self.probe_length,
                    probe_stride=self.probe_stride,
                    allow_small_seqs=self.allow_small_seqs,
                    seq_length_to_skip=self.seq_length_to_skip)
            if len(c) == 0:
                logger.warning("There are no candidate probes for a grouping "
                               "of genomes")
            cand.append(c)
        self._candidate_strs = cand
        if type(first) is DuplicateFilter:
            uniq = [list(dict.fromkeys(c)) for c in cand]
        else:   # one _filter call per group, in order, like BaseFilter.filter
            uniq = [first._filter_strs(c) for c in cand]
        ids = scf._filter_strs(uniq, self.genomes, assume_unique=True)
        chosen = [[u[i] for i in sel] for u, sel in zip(uniq, ids)]
        self.final_probes = [probe.Probe.from_str(s) for s in
                             dict.fromkeys(itertools.chain(*chosen))]

    @property
    def candidate_probes(self):
        if self._candidates is None and self._candidate_strs is not None:
            self._candidates = [probe.Probe.from_str(s) for s in
                                itertools.chain(*self._candidate_strs)]
        return self._candidates

    @candidate_probes.setter
    def candidate_probes(self, value):
        self._candidates = value

    def design(self):
        if (len(self.filters) == 2 and type(self.filters[0]) in (
                DuplicateFilter, NearDuplicateFilterWithHammingDistance,
                NearDuplicateFilterWithMinHash)
                and type(self.filters[1]) is SetCoverFilter):
            return self._design_on_strings()
        candidates, probes = self._design_for_genomes(self.genomes,
                                                      self.filters)
        self.candidate_probes = list(itertools.chain(*candidates))
        # the reference takes list(set(...)) (CPython set order); a stable
        # order-preserving de-duplication gives the same set reproducibly
        self.final_probes = list(dict.fromkeys(itertools.chain(*probes)))
