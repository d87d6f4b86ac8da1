"""ProbeDesigner: candidate probes per group (or per cluster of sequences), then
the filter list over grouped input (mirrors catch/filter/probe_designer.py:
_cluster_genomes :78-184, _pass_through_filters :186-228, _design_for_genomes
:230-271, design :273-315)."""
import itertools
import logging

import numpy as np

from catch_amd import _lib
from catch_amd import engine
from catch_amd import genome
from catch_amd import probe
from catch_amd.filter import candidate_probes
from catch_amd.filter.duplicate_filter import DuplicateFilter
from catch_amd.filter.near_duplicate_filter import (
    NearDuplicateFilterWithHammingDistance, NearDuplicateFilterWithMinHash)
from catch_amd.filter.set_cover_filter import SetCoverFilter
from catch_amd.utils import cluster

logger = logging.getLogger(__name__)


class ClusteredFragments:
    """The clusters of a clustered design as the reference returns them -- a list of lists of single-sequence Genomes
    (catch/filter/probe_designer.py:78-184) -- over fragments that are still VIEWS of their parents' storage
    (engine.FragmentTable).  The device front end reads the views (`table`, `clusters`: index arrays into it) and never
    builds a Genome or a str per fragment (224 k of them at configs[4]: 0.6 s of the step); everything else indexes or
    iterates the object like the list it stands for, and gets real Genome objects made on first use."""

    def __init__(self, table, clusters):
        self.table = table
        self.clusters = [np.asarray(c, dtype=np.int64) for c in clusters]
        self._made = {}

    def __len__(self):
        return len(self.clusters)

    def __getitem__(self, gi):
        if isinstance(gi, slice):
            return [self[i] for i in range(*gi.indices(len(self)))]
        if gi < 0:
            gi += len(self)
        got = self._made.get(gi)
        if got is None:
            one = genome.Genome.from_one_seq
            got = self._made[gi] = [one(self.table.string(int(i))) for i in self.clusters[gi]]
        return got

    def __iter__(self):
        return (self[gi] for gi in range(len(self)))

    def group_bases(self):
        """Bases of every cluster (sum of Genome.size() over its members)."""
        ln = self.table.length
        return np.array([int(ln[c].sum()) for c in self.clusters], dtype=np.int64)


class ProbeDesigner:
    def __init__(self, genomes, filters, probe_length, probe_stride, allow_small_seqs=None,
                 seq_length_to_skip=None, cluster_threshold=None, cluster_merge_after=None,
                 cluster_method=None, cluster_fragment_length=None):
        # the reference's constructor arguments, kept under the reference's attribute names
        # (catch/filter/probe_designer.py:34-76)
        given = dict(locals())
        for name in ("genomes", "filters", "probe_length", "probe_stride", "allow_small_seqs",
                     "seq_length_to_skip", "cluster_threshold", "cluster_merge_after", "cluster_method",
                     "cluster_fragment_length"):
            setattr(self, name, given[name])
        self._candidates = None
        self._candidate_strs = None
        self.final_probes = None

    # -- clustering pre-step ------------------------------------------------
    def _sequences_to_cluster(self):
        """Every sequence of every genome of every group, cut into fragments
        when asked to, short ones skipped -- in input order, which is the order
        the clustering numbers them in."""
        out = []
        L, skip = self.cluster_fragment_length, self.seq_length_to_skip
        for grp in self.genomes:
            for g in grp:
                for seq in g.seqs:
                    n = len(seq)
                    if L is None or 0 < n <= L:
                        pieces = (seq,)      # (what Genome.break_into_fragments(L, include_full_end=True) makes of it,
                    else:                    #  without a Genome and an OrderedDict per input genome: 198,838 of them)
                        pieces = [seq[i:i + L] if i + L <= n else seq[max(0, n - L):] for i in range(0, n, L)]
                    for s_ in pieces:
                        if skip is None or len(s_) > skip:
                            out.append(s_)
        return out

    def _fragment_table(self):
        """_sequences_to_cluster as views (engine.FragmentTable): the same fragments in the same order, none of them
        sliced out -- or None when some sequence is not a plain-ASCII str (the caller then slices)."""
        L, skip = self.cluster_fragment_length, self.seq_length_to_skip
        parents = [seq for grp in self.genomes for g in grp for seq in g.seqs]
        n = np.fromiter(map(len, parents), dtype=np.int64, count=len(parents))
        if L is None:
            npieces = np.ones(n.size, dtype=np.int64)
        else:
            # (0 < n <= L: the sequence itself; else ceil(n / L) pieces -- none of an empty sequence --, the last one
            # moved back to be L long)
            npieces = np.where((n > 0) & (n <= L), 1, -(-n // L))
        par = np.repeat(np.arange(n.size, dtype=np.int64), npieces)
        first = np.cumsum(npieces) - npieces
        j = np.arange(par.size, dtype=np.int64) - first[par]          # piece number inside its parent
        if L is None:
            st, ln = np.zeros(par.size, dtype=np.int64), n[par]
        else:
            whole = (n[par] <= L)
            st = np.where(whole, 0, np.minimum(j * L, np.maximum(n[par] - L, 0)))
            ln = np.where(whole, n[par], L)
        if skip is not None:
            keep = ln > skip
            par, st, ln = par[keep], st[keep], ln[keep]
        return engine.FragmentTable.build(parents, par, st, ln)

    def _resolve_cluster_method(self):
        """'choose' means connected components unless whole long genomes were
        cut into fragments: their pieces would chain into one giant component,
        which average linkage avoids (catch/filter/probe_designer.py:113-158)."""
        if self.cluster_method != "choose":
            return self.cluster_method
        if self.cluster_fragment_length is None:
            return "simple"
        sizes = [(len(g.seqs), g.size()) for grp in self.genomes for g in grp]
        nseq, total = sum(n for n, _ in sizes), sum(t for _, t in sizes)
        long_genomes = nseq > 1 and total / nseq > self.cluster_fragment_length
        return "hierarchical" if long_genomes else "simple"

    def _cluster_genomes(self):
        """One list of single-sequence Genomes per cluster of the MinHash
        clustering, largest cluster first (probe_designer.py:78-184)."""
        if len(self.genomes) > 1:
            logger.warning("Clustering ignores the %d input groupings: anything that "
                           "relies on them (e.g. --identify) no longer sees them",
                           len(self.genomes))
        import time
        t0 = time.perf_counter()
        table = None if _lib.test_env("CATCHHIP_CLUSTER_SLICE_FRAGMENTS") else self._fragment_table()
        if table is not None:
            # (round 6) fragments as views of the genomes' own storage, clusters as index arrays: no str, no Genome
            # per fragment unless somebody asks for one
            t1 = time.perf_counter()
            method = self._resolve_cluster_method()
            logger.info("MinHash clustering of %d sequences (%s, threshold %f)", len(table), method, self.cluster_threshold)
            clusters = cluster.cluster_with_minhash_signatures(table, threshold=self.cluster_threshold, cluster_method=method)
            logger.info("%d clusters; sizes %s", len(clusters), [len(c) for c in clusters])
            t2 = time.perf_counter()
            out = ClusteredFragments(table, clusters)
            self.cluster_timings = dict(cluster.last_timings, fragments_s=t1 - t0, genomes_s=time.perf_counter() - t2)
            return out
        seqs = self._sequences_to_cluster()
        t1 = time.perf_counter()
        method = self._resolve_cluster_method()
        logger.info("MinHash clustering of %d sequences (%s, threshold %f)",
                    len(seqs), method, self.cluster_threshold)
        clusters = cluster.cluster_with_minhash_signatures(
            dict(enumerate(seqs)), threshold=self.cluster_threshold,
            cluster_method=method)
        logger.info("%d clusters; sizes %s", len(clusters), [len(c) for c in clusters])
        t2 = time.perf_counter()
        one = genome.Genome.from_one_seq
        out = [[one(seqs[i]) for i in members] for members in clusters]
        # wall seconds by stage of the clustering pre-step (tools/s5_time.py, bench.py --workload S5)
        self.cluster_timings = dict(cluster.last_timings, fragments_s=t1 - t0, genomes_s=time.perf_counter() - t2)
        return out

    # -- object pipeline ----------------------------------------------------
    @staticmethod
    def _run_filters(filters, probes, genomes, grouped):
        for f in filters:
            logger.info("Filter %s", type(f).__name__)
            probes = f.filter(probes, genomes, input_is_grouped=grouped)
        return probes

    def _pass_through_filters_ungrouped(self, probes, genomes, filters):
        return self._run_filters(filters, probes, genomes, False)

    def _pass_through_filters(self, probes, genomes, filters):
        if len(probes) != len(genomes):
            raise ValueError("one list of probes per group of genomes")
        return self._run_filters(filters, probes, genomes, True)

    def _candidates_of_group(self, grp):
        seqs = [s for g in grp for s in g.seqs]
        if not seqs:
            return []
        return candidate_probes.make_candidate_probes_from_sequences(
            seqs, self.probe_length, self.probe_stride, **self._window_options())

    def _window_options(self, small=True):
        """The keyword arguments of the candidate generators."""
        opts = {"seq_length_to_skip": self.seq_length_to_skip}
        if small:
            opts["allow_small_seqs"] = self.allow_small_seqs
        return opts

    def _design_for_genomes(self, genomes, filters):
        """(candidates per group, what the filters leave of them)."""
        candidates = [self._candidates_of_group(grp) for grp in genomes]
        for c in candidates:
            if not c:
                logger.warning("A group of genomes has no candidate probes")
        return candidates, self._pass_through_filters(candidates, genomes, filters)

    def _design_on_strings(self, genomes, filters):
        """[DuplicateFilter | near-duplicate filter, SetCoverFilter] -- the
        filter lists bin/design.py:296-340 builds -- on plain strings:
        candidates are sliced, de-duplicated (dict, or the LSH filter on the
        device) and handed to the set cover filter without a Probe object per
        candidate (a design over 8,000 genomes spent 0.85 of its 1.0 s building
        them); only the selected probes become objects."""
        first, scf = filters
        mode = self._device_front_end_mode(genomes, first, scf)
        if mode is not None:
            self._candidate_strs = None
            self._candidate_genomes = genomes
            run = (scf._filter_genomes_device if mode == "per group"
                   else scf._filter_genomes_device_union)
            chosen = run(genomes, self.probe_length, self.probe_stride,
                         self.seq_length_to_skip,
                         None if type(first) is DuplicateFilter else first)
            return [[probe.Probe.from_str(s) for s in grp] for grp in chosen]
        cand = []
        for grp in genomes:
            c = []
            for g in grp:
                c += candidate_probes.candidate_strings_from_sequences(
                    list(g.seqs), self.probe_length, self.probe_stride, **self._window_options())
            if len(c) == 0:
                logger.warning("There are no candidate probes for a grouping "
                               "of genomes")
            cand.append(c)
        self._candidate_strs = cand
        if type(first) is DuplicateFilter:
            uniq = [list(dict.fromkeys(c)) for c in cand]
        elif hasattr(first, "_filter_strs_many"):
            uniq = first._filter_strs_many(cand)
        else:   # one _filter call per group, in order, like BaseFilter.filter
            uniq = [first._filter_strs(c) for c in cand]
        ids = scf._filter_strs(uniq, genomes, assume_unique=True)
        chosen = [[u[i] for i in sel] for u, sel in zip(uniq, ids)]
        return [[probe.Probe.from_str(s) for s in grp] for grp in chosen]

    def _device_front_end_mode(self, genomes, first, scf):
        """Candidates and the duplicate (or near-duplicate) filter on the device:
        one of the usual filter pairs without ranks, no --small-seq-min, every
        sequence a str at least a probe long (or skipped).  "per group": few or
        large groups, each its own instance; "union": many small groups
        (clusters) as one instance with group numbers; None: host front end."""
        import os
        if os.environ.get("CATCHHIP_HOST_FRONT_END"):
            return None
        if self.allow_small_seqs:
            return None
        if type(first) is NearDuplicateFilterWithHammingDistance:
            if first.dim != self.probe_length:
                return None              # the host path raises the reference's error
        elif type(first) is NearDuplicateFilterWithMinHash:
            if not (first.kmer_size <= self.probe_length <= first.kmer_size + 255):
                return None
        elif type(first) is not DuplicateFilter:
            return None
        if scf.identify or scf.avoided_genomes:
            return None
        skip, L = self.seq_length_to_skip, self.probe_length
        total, ngroups = 0, 0
        if isinstance(genomes, ClusteredFragments):
            # (views of plain-ASCII strs: the same conditions on the lengths, from the table)
            ln = genomes.table.length
            used = np.concatenate(genomes.clusters) if len(genomes) else np.zeros(0, dtype=np.int64)
            ln = ln[used]
            if skip is not None:
                ln = ln[ln > skip]
            if ln.size and int(ln.min()) < L:
                return None               # the host path raises the reference's error
            total = int(ln.sum())
            ngroups = sum(1 for c in genomes.clusters if len(c))
            genomes = ()
        for grp in genomes:
            if len(grp) == 0:
                continue
            ngroups += 1
            for g in grp:
                for s in g.seqs:
                    if not isinstance(s, str):
                        return None
                    n = len(s)
                    if skip is not None and n <= skip:
                        continue
                    if n < L:
                        return None       # the host path raises the reference's error
                    total += n
        if ngroups == 0:
            return None
        # per group: few groups, or groups large enough to fill the GPU on their own (S4: 30 Mbases
        # each).  Many clusters of a clustered design go through union instances whatever their size:
        # a set cover over one long genome is a chain of ties (its picks come one per round), and in a
        # union the chains of all clusters advance in the same rounds (S5 x 0.25, 2,293 clusters of
        # 0.4 Mbases: 164,418 rounds one cluster at a time)
        if ngroups < 8 or total >= int(_lib.test_env("CATCHHIP_UNION_MAX_MEAN_BASES", "4000000")) * ngroups:
            return "per group"
        return "union"

    @staticmethod
    def _strings_path_ok(filters):
        return (len(filters) >= 2 and type(filters[0]) in (
            DuplicateFilter, NearDuplicateFilterWithHammingDistance,
            NearDuplicateFilterWithMinHash)
            and type(filters[1]) is SetCoverFilter)

    @property
    def candidate_probes(self):
        if (self._candidates is None and self._candidate_strs is None
                and getattr(self, "_candidate_genomes", None) is not None):
            # the device front end never built them: do it now, on request
            self._candidate_strs = [
                [s for g in grp for s in candidate_probes.candidate_strings_from_sequences(
                    list(g.seqs), self.probe_length, self.probe_stride, **self._window_options(small=False))]
                for grp in self._candidate_genomes]
        if self._candidates is None and self._candidate_strs is not None:
            self._candidates = [probe.Probe.from_str(s) for s in
                                itertools.chain(*self._candidate_strs)]
        return self._candidates

    @candidate_probes.setter
    def candidate_probes(self, value):
        self._candidates = value

    def design(self):
        clustered = self.cluster_threshold is not None
        if not clustered:
            genomes, before, after = self.genomes, self.filters, []
        if clustered:
            # design per cluster up to cluster_merge_after, then run the
            # remaining filters on the merged probes (:291-315)
            assert self.cluster_merge_after is not None and self.cluster_merge_after in self.filters
            merge_idx = self.filters.index(self.cluster_merge_after) + 1
            before, after = self.filters[:merge_idx], self.filters[merge_idx:]
            genomes = self._cluster_genomes()
        if self._strings_path_ok(before):
            # the two expensive filters on strings, any later ones (adapters)
            # on the few selected probes as objects
            grouped = self._design_on_strings(genomes, before[:2])
            grouped = self._pass_through_filters(grouped, genomes, before[2:])
            probes = list(dict.fromkeys(itertools.chain(*grouped)))
        else:
            candidates, grouped = self._design_for_genomes(genomes, before)
            self.candidate_probes = list(itertools.chain(*candidates))
            # the reference takes list(set(...)) (CPython set order); a stable
            # order-preserving de-duplication gives the same set reproducibly
            probes = list(dict.fromkeys(itertools.chain(*grouped)))
        if after:
            probes = self._pass_through_filters_ungrouped(probes, genomes,
                                                          after)
        self.final_probes = probes
