"""SetCoverFilter on the MI355X: drop-in for catch/filter/set_cover_filter.py.

Same constructor and `BaseFilter.filter()` contract as the reference class
(catch/filter/set_cover_filter.py:199-357, :902-930); the work of
`_make_sets` (:359-470), `_make_ranks` (:614-735) and
`set_cover.approx_multiuniverse` (catch/utils/set_cover.py:147-615) runs in
the HIP kernels of libcatchhip.so through `catch_amd.engine`.  There is no
CPU fallback: without the library or a GPU the filter raises.

Not supported (raises NotImplementedError): custom hybridization functions
loaded from a Python file (`custom_cover_range_fn`, :288-299) -- an arbitrary
Python callable cannot run inside a kernel.
"""
import logging
import os

import numpy as np

from catch_amd import _lib
from catch_amd import engine
from catch_amd import probe
from catch_amd.filter.base_filter import BaseFilter
from catch_amd.utils import seq_io

logger = logging.getLogger(__name__)


def set_max_num_processes_for_set_cover_instances(max_num_processes=8):
    """catch/filter/set_cover_filter.py:66-79.  Accepted for CLI
    compatibility; set-cover instances are solved on the GPU."""
    global _sc_max_num_processes
    _sc_max_num_processes = max_num_processes


set_max_num_processes_for_set_cover_instances()

_RC = str.maketrans("ACGT", "TGCA")


def _reverse_complement(s):
    """A<->T, C<->G, everything else maps to itself (:515-521)."""
    return s[::-1].translate(_RC)


class SetCoverFilter(BaseFilter):
    """Selects candidate probes with the greedy multi-universe set cover."""

    def __init__(self,
                 mismatches,
                 lcf_thres,
                 island_of_exact_match=0,
                 mismatches_tolerant=None,
                 lcf_thres_tolerant=None,
                 island_of_exact_match_tolerant=None,
                 custom_cover_range_fn=None,
                 custom_cover_range_tolerant_fn=None,
                 identify=False,
                 avoided_genomes=[],
                 coverage=1.0,
                 cover_extension=0,
                 kmer_probe_map_k=20,
                 kmer_probe_map_use_native_dict=False):
        if (custom_cover_range_fn is not None
                or custom_cover_range_tolerant_fn is not None):
            raise NotImplementedError(
                "custom hybridization functions cannot run on the GPU path")
        # The seed scan (cover threshold == probe length, the default) takes any
        # mismatch budget; the general path (-l below the probe length, island,
        # unequal probes, other alphabets) keeps the nearest mismatches of each
        # side in fixed arrays of 32 (csrc/scan.hip MAX_MM)
        for name, val in (("mismatches", mismatches),
                          ("mismatches_tolerant", mismatches_tolerant)):
            if val is not None and val > 32 and lcf_thres is not None:
                logger.warning("%s = %d: cover ranges below the full probe length "
                               "(-l / --island-of-exact-match) support at most 32 "
                               "mismatches on the GPU and will raise", name, val)
        self.mismatches = mismatches
        self.lcf_thres = lcf_thres
        self.island_of_exact_match = island_of_exact_match
        # tolerant parameters default to the strict ones when falsy (:308-313)
        if not mismatches_tolerant:
            mismatches_tolerant = mismatches
        if not lcf_thres_tolerant:
            lcf_thres_tolerant = lcf_thres
        if not island_of_exact_match_tolerant:
            island_of_exact_match_tolerant = island_of_exact_match
        self.mismatches_tolerant = mismatches_tolerant
        self.lcf_thres_tolerant = lcf_thres_tolerant
        self.island_of_exact_match_tolerant = island_of_exact_match_tolerant

        if identify:
            if (coverage <= 1.0 and coverage >= 0.25) or \
               (coverage > 1 and coverage >= 5000):
                logger.warning(("Identification is enabled but the required "
                                "coverage is high; generally coverage should "
                                "be small when performing identification"))
        self.identify = identify
        self.avoided_genomes = avoided_genomes
        self.coverage = coverage
        self.cover_extension = cover_extension
        self.kmer_probe_map_k = kmer_probe_map_k
        self.kmer_probe_map_use_native_dict = kmer_probe_map_use_native_dict
        self.requires_probe_groupings = True
        self._force_num_processes = None   # accepted, unused (:355-357)
        self.scan_mode = engine.SCAN_AUTO
        self.last_timings = {}

    # ------------------------------------------------------------------
    def _context(self):
        return engine.default_context()

    def _make_sets(self, candidate_probes, target_genomes, ctx=None,
                   targets=None):
        """Device rows for one group (:359-470).  Returns (rows, owner) where
        rows is an engine.Rows (set ids = indices into candidate_probes)."""
        ctx = ctx or self._context()
        strs = [p.seq_str for p in candidate_probes]
        k, uniq, owner, ep, eo = probe.anchor_table(
            strs, self.mismatches, self.lcf_thres,
            min_k=self.kmer_probe_map_k, k=self.kmer_probe_map_k)
        own_targets = targets is None
        if own_targets:
            targets = engine.Targets(ctx, [g.seqs for g in target_genomes])
        probes = engine.Probes(ctx, uniq, owner, ep, eo, k)
        try:
            rows = engine.Rows.scan(ctx, probes, targets, self.mismatches,
                                    self.lcf_thres,
                                    self.island_of_exact_match,
                                    self.cover_extension, self.scan_mode)
        finally:
            probes.close()
            if own_targets:
                targets.close()
        return rows

    def _tolerant_bp_over(self, ctx, probes_dev, n_uniq, sequences):
        """Sum over `sequences` and their reverse complements of the bp each
        unique probe covers under the tolerant model (:472-529)."""
        bp = np.zeros(max(n_uniq, 1), dtype=np.int64)
        if not sequences:
            return bp
        # one genome per sequence: coverage is merged per sequence anyway
        genomes = []
        for s in sequences:
            genomes.append([s])
            genomes.append([_reverse_complement(s)])
        t = engine.Targets(ctx, genomes)
        try:
            engine.tolerant_bp(ctx, probes_dev, t, self.mismatches_tolerant,
                               self.lcf_thres_tolerant,
                               self.island_of_exact_match_tolerant, bp)
        finally:
            t.close()
        return bp

    def _make_ranks(self, candidate_probes, target_genomes_grouped, ctx=None):
        return self._make_ranks_strs([p.seq_str for p in candidate_probes],
                                     target_genomes_grouped, ctx)

    def _make_ranks_strs(self, strs, target_genomes_grouped, ctx=None):
        """Rank per candidate (:614-735): (0, #groups hit) under --identify,
        (1, avoided bp) for probes that touch an avoided genome, densified."""
        n = len(strs)
        need = self.identify or len(self.avoided_genomes) > 0
        if not need:
            return np.zeros(n, dtype=np.int64)
        ctx = ctx or self._context()
        k, uniq, _owner, ep, eo = probe.anchor_table(
            strs, self.mismatches_tolerant, self.lcf_thres_tolerant,
            min_k=self.kmer_probe_map_k, k=self.kmer_probe_map_k)
        ident = np.arange(len(uniq), dtype=np.int32)
        pd = engine.Probes(ctx, uniq, ident, ep, eo, k)
        try:
            if self.identify:
                hits = np.zeros(len(uniq), dtype=np.int64)
                for genomes in target_genomes_grouped:
                    seqs = [s for g in genomes for s in g.seqs]
                    bp = self._tolerant_bp_over(ctx, pd, len(uniq), seqs)
                    hits += (bp[:len(uniq)] >= 1)
                rank_a = np.zeros(len(uniq), dtype=np.int64)
                rank_b = hits
            else:
                rank_a = np.zeros(len(uniq), dtype=np.int64)
                rank_b = np.zeros(len(uniq), dtype=np.int64)
            total = np.zeros(len(uniq), dtype=np.int64)
            for path in self.avoided_genomes:
                for sequence in seq_io.iterate_fasta(path):
                    total += self._tolerant_bp_over(ctx, pd, len(uniq),
                                                    [sequence])[:len(uniq)]
            avoided = total > 0
            rank_a = np.where(avoided, 1, rank_a)
            rank_b = np.where(avoided, total, rank_b)
        finally:
            pd.close()
        tuples = sorted(set(zip(rank_a.tolist(), rank_b.tolist())))
        tidx = {t: i for i, t in enumerate(tuples)}
        uidx = {s: i for i, s in enumerate(uniq)}
        return np.fromiter(
            (tidx[(int(rank_a[uidx[s]]), int(rank_b[uidx[s]]))] for s in strs),
            dtype=np.int64, count=n)

    def _make_universe_p(self, target_genomes):
        """:761-792."""
        if self.coverage <= 1.0:
            return [self.coverage for _ in target_genomes]
        out = []
        for gnm in target_genomes:
            desired = min(self.coverage, gnm.size())
            out.append(float(desired) / gnm.size())
        return out

    def _filter(self, input, target_genomes_grouped):
        """input = [p_1, ..., p_m] candidate probes per group; returns the
        selected probes per group (:902-930)."""
        input = [list(pp) for pp in input]
        ids = self._filter_strs([[p.seq_str for p in pp] for pp in input],
                                target_genomes_grouped)
        return [[pp[i] for i in sel] for pp, sel in zip(input, ids)]

    def _filter_strs(self, input_strs, target_genomes_grouped,
                     assume_unique=False, only=None, tables=None):
        """The filter on plain probe strings: per group the indices of the
        selected candidates, in pick order.  Groups are independent instances:
        up to CATCHHIP_GROUPS_IN_FLIGHT (default 4) of them run at once, each
        on its own context / HIP stream (catchhip_setcover_filter_many).
        assume_unique: the strings of a group are pairwise distinct (they come
        out of the duplicate filter).  only: restrict the work to these group
        indices (the others come back empty; multi-rank runs).  With more than
        one rank (catch_amd.parallel.init_from_env) the groups are spread over
        the ranks and every rank returns the complete selection."""
        import os
        from catch_amd import parallel
        if only is None and parallel.world().size > 1:
            return self._filter_strs_multirank(input_strs, target_genomes_grouped,
                                               assume_unique, parallel.world())
        selected = [[] for _ in input_strs]
        timings = dict(scan_ms=0.0, rows_ms=0.0, greedy_ms=0.0, picks=0,
                       rows=0, scan_launches=0, greedy_launches=0)
        nonempty = [i for i, pp in enumerate(input_strs) if len(pp) > 0]
        todo = [i for i in nonempty if only is None or i in only]
        random_anchors = any(probe.anchors_use_random(
            input_strs[i], self.mismatches, self.lcf_thres, self.kmer_probe_map_k)
            for i in nonempty)
        width = max(1, int(os.environ.get("CATCHHIP_GROUPS_IN_FLIGHT", "4")))
        # many SMALL groups go through one instance (its anchors are the groups'
        # anchors drawn back to back, in input order); large groups (thousands of
        # candidates each) overlap better as separate instances in flight
        if (only is None and tables is None
                and assume_unique and not self.identify and not self.avoided_genomes
                and len(todo) >= int(_lib.test_env("CATCHHIP_UNION_MIN_GROUPS", "8"))
                and sum(len(input_strs[gi]) for gi in todo) <= len(todo) * int(
                    _lib.test_env("CATCHHIP_UNION_MAX_MEAN_CANDIDATES", "8192"))
                and len({len(s) for gi in todo for s in input_strs[gi]}) == 1):
            self._filter_strs_union(input_strs, target_genomes_grouped, todo,
                                    selected, timings)
            todo = []
        # Random anchors (catch/probe.py:391-401) consume np.random once per
        # probe, group after group in INPUT order, while the groups below run
        # largest first (and, over several ranks, only some of them here): the
        # tables of all groups are then drawn up front, in input order, so that
        # every schedule selects what the reference selects.
        # (With --identify / --avoid-genomes the rank tables draw too, interleaved
        # with the set tables group by group: then the groups simply keep their
        # input order.)
        need_ranks = self.identify or len(self.avoided_genomes) > 0
        in_input_order = random_anchors and need_ranks and tables is None
        if todo and random_anchors and tables is None and not need_ranks:
            tables = self._anchor_tables_in_input_order(input_strs, nonempty,
                                                        assume_unique)
        if in_input_order:
            chunks = [todo[i:i + width] for i in range(0, len(todo), width)]
        else:
            # largest first, as the reference hands its instances to the pool
            # (catch/filter/set_cover_filter.py:880-887)
            todo.sort(key=lambda i: (-len(input_strs[i]), i))
            chunks = _chunks_by_size(
                todo, lambda gi: sum(g.size() for g in target_genomes_grouped[gi]), width)
        for chunk in chunks:
            ctxs = _contexts(len(chunk))
            specs, held, all_ranks = [], [], []
            try:
                for ctx, gi in zip(ctxs, chunk):
                    strs, target_genomes = \
                        input_strs[gi], target_genomes_grouped[gi]
                    logger.info("Building set cover sets input (group %d of %d)",
                                gi + 1, len(input_strs))
                    k, uniq, owner, ep, eo = (
                        tables.pop(gi) if tables is not None else
                        probe.anchor_table(
                            strs, self.mismatches, self.lcf_thres,
                            min_k=self.kmer_probe_map_k, k=self.kmer_probe_map_k,
                            assume_unique=assume_unique))
                    targets = engine.Targets(ctx, [g.seqs for g in target_genomes])
                    held.append(targets)
                    probes = engine.Probes(ctx, uniq, owner, ep, eo, k)
                    held.append(probes)
                    ranks = self._make_ranks_strs(strs, target_genomes_grouped,
                                                  ctx)
                    all_ranks.append(ranks)
                    specs.append((ctx, probes, targets, len(strs),
                                  ranks if ranks.any() else None,
                                  self._make_universe_p(target_genomes)))
                logger.info("Solving set cover instances (groups %s of %d)",
                            [gi + 1 for gi in chunk], len(input_strs))
                results = engine.setcover_filter_many(
                    specs, self.mismatches, self.lcf_thres,
                    self.island_of_exact_match, self.cover_extension,
                    self.scan_mode)
            finally:
                for h in held:
                    h.close()
            for ctx, gi, ranks, (ids, nrows) in zip(ctxs, chunk, all_ranks,
                                                    results):
                _accumulate(timings, ctx, nrows, len(ids))
                num_bad = int(np.count_nonzero(ranks[ids] > 0)) if len(ids) else 0
                if num_bad > 0:
                    logger.warning(("Group %d: forced to choose %d less-than-ideal "
                                    "probe%s (i.e., probes that 'hit' more than "
                                    "one grouping during identification or probes "
                                    "that cover an avoided genome)"), gi + 1,
                                   num_bad, "" if num_bad == 1 else "s")
                selected[gi] = list(ids)
        self.last_timings = timings
        return selected


    def _anchor_tables_in_input_order(self, input_strs, nonempty, assume_unique):
        """{group: anchor table} for all non-empty groups, drawn in input order,
        when the anchors are random; None when they are the pigeonhole anchors
        (no random numbers: tables are then built where they are used)."""
        if not any(probe.anchors_use_random(input_strs[i], self.mismatches,
                                            self.lcf_thres, self.kmer_probe_map_k)
                   for i in nonempty):
            return None
        return {i: probe.anchor_table(input_strs[i], self.mismatches, self.lcf_thres,
                                      min_k=self.kmer_probe_map_k,
                                      k=self.kmer_probe_map_k,
                                      assume_unique=assume_unique)
                for i in nonempty}

    def _filter_strs_multirank(self, input_strs, target_genomes_grouped,
                               assume_unique, W):
        """The filter over W.size ranks (one process per GPU).  Level 1: whole
        groups go to ranks longest first (the reference's pool order,
        catch/filter/set_cover_filter.py:880-887), no collective.  Level 2: a
        group above an even share of the bases is sharded over ALL ranks by
        universes: every rank scans the group's candidates against its own
        range of genomes and the frontier solver's rounds exchange the
        per-candidate gains (RCCL all-reduce, catch_amd/parallel.py).  Sharding
        needs full coverage; groups under partial coverage stay whole.  Ranks
        (identify / avoided genomes) are computed by every rank for the groups
        it shares.  Every rank returns every group's selection."""
        import os
        from catch_amd import parallel
        n = len(input_strs)
        if ((self.identify or self.avoided_genomes) and any(
                probe.anchors_use_random(input_strs[i], self.mismatches, self.lcf_thres,
                                         self.kmer_probe_map_k) for i in range(n))):
            # random anchors for the set AND the rank tables, interleaved group by
            # group: every rank does every group, in input order
            return self._filter_strs(input_strs, target_genomes_grouped, assume_unique,
                                     only=set(range(n)))
        costs = [sum(g.size() for g in target_genomes_grouped[i]) if len(input_strs[i]) else 0
                 for i in range(n)]
        # (ranks -- identify / avoided genomes -- and partial coverage are sharded too since round 4: every rank computes
        # the group's ranks itself, they only gate which sets may claim; need[u] and the acceptance thresholds of a
        # universe live on the rank that owns it, the candidates' verdicts travel with the lost marks)
        eligible = True
        # (a group with fewer genomes than ranks is not sharded: some rank would hold an empty shard)
        sharded, whole = parallel.plan_with_sharding(
            costs, W.size,
            min_cost=int(os.environ.get("CATCHHIP_SHARD_MIN_BASES", "30000000"))
            if eligible else float("inf"),
            eligible=[len(target_genomes_grouped[i]) >= W.size for i in range(n)])
        selected = [[] for _ in range(n)]
        timings = dict(scan_ms=0.0, rows_ms=0.0, greedy_ms=0.0, picks=0, rows=0,
                       scan_launches=0, greedy_launches=0, sharded_groups=list(sharded))
        ctx = W.comm_ctx
        fallback = []
        nonempty = [i for i in range(n) if len(input_strs[i]) > 0]
        # Random anchors: every rank must scan a sharded group with the SAME tables, and the
        # selection must be the single-rank one: rank 0's np.random state goes to every rank before
        # the tables of all groups are drawn (in input order, as one rank draws them)
        if any(probe.anchors_use_random(input_strs[i], self.mismatches, self.lcf_thres,
                                        self.kmer_probe_map_k) for i in nonempty):
            np.random.set_state(W.broadcast(np.random.get_state()))
        tables = self._anchor_tables_in_input_order(input_strs, nonempty, assume_unique)
        for gi in sharded:                       # all ranks together, same order
            strs, genomes = input_strs[gi], target_genomes_grouped[gi]
            logger.info("Set cover of group %d of %d sharded over %d ranks",
                        gi + 1, n, W.size)
            rows = shard = probes = targets = None
            err, qualifies = None, False
            try:
                # a failure on one rank (out of memory, a library error) is reported to ALL ranks below:
                # nobody may walk into the solver's all-reduces alone
                try:
                    k, uniq, owner, ep, eo = (
                        tables[gi] if tables is not None else
                        probe.anchor_table(
                            strs, self.mismatches, self.lcf_thres,
                            min_k=self.kmer_probe_map_k, k=self.kmer_probe_map_k,
                            assume_unique=assume_unique))
                    b = parallel.split_universes([g.size() for g in genomes], W.size)
                    targets = engine.Targets(ctx, [g.seqs for g in genomes[b[W.rank]:b[W.rank + 1]]])
                    probes = engine.Probes(ctx, uniq, owner, ep, eo, k)
                    rows = engine.Rows.scan(ctx, probes, targets, self.mismatches,
                                            self.lcf_thres, self.island_of_exact_match,
                                            self.cover_extension, self.scan_mode)
                    ranks = (self._make_ranks_strs(strs, target_genomes_grouped, ctx)
                             if (self.identify or self.avoided_genomes) else None)
                    try:
                        up_all = self._make_universe_p(genomes)
                        part = any(p_ < 1.0 for p_ in up_all)     # (the same on every rank)
                        # partial-ness is a property of the INSTANCE: a rank whose own genomes all have p == 1
                        # (coverage in bases, genomes shorter than it) still builds a partial shard -- the same
                        # kernels, round shape and refusals on every rank
                        shard = engine.Shard(rows, len(strs), ranks,
                                             up_all[b[W.rank]:b[W.rank + 1]] if part else None,
                                             instance_partial=part)
                        qualifies = True
                    except ValueError as exc:
                        # the expected refusals: rows too long for the sharded kernels, partial coverage with too few
                        # sets for the row-parallel ones -> whole group
                        if "longer than 257" not in str(exc) and "row-parallel kernels only" not in str(exc):
                            raise
                except Exception as exc:          # noqa: BLE001 -- reported collectively
                    err = "%s: %s" % (type(exc).__name__, exc)
                status = W.allgather((err, qualifies))
                W_errs = [(r, e) for r, (e, _q) in enumerate(status) if e]
                if W_errs:
                    raise RuntimeError("sharded set cover of group %d failed: %s" % (
                        gi + 1, "; ".join("rank %d: %s" % re for re in W_errs)))
                # every rank must take the same path
                if all(q for _e, q in status):
                    selected[gi] = parallel.sharded_solve([shard], W.exchange_for([shard]), W.native_for([shard]))
                    timings["rows"] += rows.n
                    timings["picks"] += len(selected[gi])
                else:
                    fallback.append(gi)
            finally:
                for h in (shard, rows, probes, targets):
                    if h is not None:
                        h.close()
        # groups that could not be sharded after all go whole to the least loaded rank
        loads = [sum(costs[i] for i in w) for w in whole]
        for gi in fallback:
            r = min(range(W.size), key=lambda q: (loads[q], q))
            whole[r].append(gi)
            loads[r] += costs[gi]
        err, mine = None, None
        try:
            mine = self._filter_strs(input_strs, target_genomes_grouped, assume_unique,
                                     only=set(whole[W.rank]), tables=tables)
        except Exception as exc:                  # noqa: BLE001 -- reported collectively
            err = "%s: %s" % (type(exc).__name__, exc)
        W.agree(err)                              # a rank that failed must not leave the others in the gather
        for k_, v in self.last_timings.items():
            if isinstance(v, (int, float)):
                timings[k_] = timings.get(k_, 0) + v
        parts = W.allgather({gi: mine[gi] for gi in whole[W.rank]})
        for part in parts:
            for gi, ids in part.items():
                selected[gi] = ids
        self.last_timings = timings
        return selected

    def _filter_genomes_device(self, target_genomes_grouped, probe_length,
                               probe_stride, seq_length_to_skip=None,
                               near_duplicate_filter=None, return_ids=False):
        """[DuplicateFilter | near-duplicate filter, SetCoverFilter] with the
        front end on the device (near_duplicate_filter: an LSH filter object to
        apply to the unique candidates, in their multiplicity order, before the
        set cover):
        per group the candidate windows of its genomes are enumerated and
        de-duplicated on the GPU (catchhip_candidates_create), gathered into a
        probes object and solved; only the selected candidates are looked up
        again, as slices of the host's sequence strings.  Returns the selected
        probe strings per group, in pick order (return_ids: the ids of the
        selected unique candidates instead -- what bench.py's M2 figure times:
        pack + H2D + front end + scan + solve + ids out).  No ranks (identify /
        avoided genomes need every candidate's string)."""
        import os
        assert not self.identify and not self.avoided_genomes
        out = [[] for _ in target_genomes_grouped]
        timings = dict(scan_ms=0.0, rows_ms=0.0, greedy_ms=0.0, picks=0,
                       rows=0, scan_launches=0, greedy_launches=0,
                       candidates=0, unique_candidates=0)
        import time as _time
        events, t_call = [], _time.perf_counter()      # (stage, item, start, end): the pipeline's timeline
        todo = [i for i, g in enumerate(target_genomes_grouped) if len(g) > 0]
        width = max(1, int(os.environ.get("CATCHHIP_GROUPS_IN_FLIGHT", "4")))
        sizes = {i: sum(g.size() for g in target_genomes_grouped[i]) for i in todo}
        # largest first (catch/filter/set_cover_filter.py:880-887); the random
        # anchors of a group are drawn when its turn comes, so that order is kept
        # only when no random numbers are involved
        # (nor while an LSH near-duplicate filter draws its positions / hash
        # functions from `random` group after group)
        keep_input_order = not (near_duplicate_filter is None and not probe.anchors_use_random(
            ["A" * probe_length], self.mismatches, self.lcf_thres, self.kmer_probe_map_k))
        # Small groups as ONE instance (as _filter_genomes_device_union does for the clusters of a clustered
        # design): a group of a few Mbases is a chain of ~60 launches that the GPU finishes in microseconds
        # each, and ten such groups one after the other are ten chains -- their union is one (S4 with resident inputs, the 17
        # groups below 32 Mbases: 69.6 -> 60.4 ms; from host strings with three lanes the groups below 16 Mbases:
        # 0.182 -> 0.178 s per pass, while a union of 247 Mbases unbalances the lanes: 0.30 s -- hence the default;
        # every group's picks and their order are its own: test_union_of_groups_equals_per_group).  Only without random numbers in the building of a group
        # (the draws of a union would come in another order than group by group).  An "item" below is a list of
        # group numbers: one group, or the groups of a union.
        union_below = int(float(os.environ.get("CATCHHIP_UNION_SMALL_BELOW_MBASES", "16")) * 1e6)
        items = {}
        if not keep_input_order and union_below > 0:
            small = [i for i in todo if sizes[i] < union_below]
            at = 0
            while len(small) - at >= 2:
                take, b = [], 0
                while at < len(small) and (not take or b + sizes[small[at]] <= 300_000_000):
                    take.append(small[at])
                    b += sizes[small[at]]
                    at += 1
                if len(take) < 2:
                    break
                key = take[0]                         # an item goes by its first group's number
                items[key] = take
                sizes[key] = b
                todo = [i for i in todo if i not in take[1:]]
        for i in todo:
            items.setdefault(i, [i])
        if not keep_input_order:
            todo.sort(key=lambda i: (-sizes[i], i))
            chunks = _chunks_by_size(todo, lambda gi: sizes[gi], width)
        else:
            chunks = [todo[i:i + width] for i in range(0, len(todo), width)]
        # The inputs of the NEXT groups are packed and uploaded while the current
        # chunk is scanned and solved: a helper thread builds (targets, candidates,
        # probes) on the upload context -- host gather into pinned memory, H2D,
        # the front-end kernels on that context's stream -- in the order the
        # chunks consume them (so `random` / np.random are drawn from in the same
        # order as without the overlap), at most CATCHHIP_PREFETCH_DEPTH groups
        # ahead; the finished objects change hands (engine.*.rebind).
        depth = int(os.environ.get("CATCHHIP_PREFETCH_DEPTH", "2"))
        order = [gi for chunk in chunks for gi in chunk]

        def genomes_of(gi):
            return [g for member in items[gi] for g in target_genomes_grouped[member]]

        def universe_p_of(gi):
            return [p for member in items[gi] for p in self._make_universe_p(target_genomes_grouped[member])]

        def build(gi, ctx=None):
            uctx = ctx or engine.upload_context()
            target_genomes = genomes_of(gi)
            made = []
            t_b0 = _time.perf_counter()
            try:
                targets = engine.Targets(uctx, [g.seqs for g in target_genomes])
                made.append(targets)
                events.append(("pack", gi, t_b0 - t_call, _time.perf_counter() - t_call))
                if len(items[gi]) > 1:
                    targets.set_groups(np.repeat(np.arange(len(items[gi])),
                                                 [len(target_genomes_grouped[member]) for member in items[gi]]))
                cands = engine.Candidates(uctx, targets, probe_length,
                                          probe_stride, seq_length_to_skip)
                made.append(cands)
                ncand, nuniq = cands.ncandidates, cands.n
                if near_duplicate_filter is not None:
                    if len(items[gi]) > 1:
                        near_duplicate_filter._apply_to_grouped_candidates(cands, len(items[gi]))
                    else:
                        near_duplicate_filter._apply_to_candidates(cands)
                probes = _probes_of_candidates(cands, _anchors_for_candidates(
                    cands.n, probe_length, self.mismatches, self.lcf_thres, self.kmer_probe_map_k))
                made.append(probes)
                events.append(("build", gi, t_b0 - t_call, _time.perf_counter() - t_call))
            except BaseException:
                for h in reversed(made):
                    h.close()
                raise
            return targets, cands, probes, ncand, nuniq

        def discard(res):
            for h in (res[2], res[1], res[0]):
                h.close()

        # Lanes (CATCHHIP_GROUP_LANES=1, the default with the prefetch on): `width` worker threads, each with
        # its own context / stream and a FIXED list of groups (longest-processing-time-first over the
        # groups' sizes, so the same group meets the same context -- and its cached device blocks -- on
        # every call); a lane starts its next group as soon as it is free: the largest group keeps one
        # lane busy while the others work through the rest, instead of chunks that wait for their slowest
        # member.  The groups are still BUILT one at a time by one thread, in an order that serves the
        # lanes as they become free (and, when random numbers are drawn while building, simply in
        # `order`: the draws must stay where they are).
        # (CATCHHIP_GROUP_LANES: 0 = off, chunks instead; 1 = three lanes -- S4 from host strings: 0.187 s per
        # pass with three, 0.192 with four, 0.205 with two, 0.215 in chunks --; n >= 2 = that many)
        lane_env = int(os.environ.get("CATCHHIP_GROUP_LANES", "1"))
        lanes = depth > 0 and lane_env != 0 and len(order) > 1
        lane_count = 3 if lane_env == 1 else lane_env
        pre = feed = None
        # (several devices, CATCHHIP_DEVICES: the chunks' contexts sit on different GPUs and an object built on the one
        # upload context cannot change hands across devices -- every group is then built on its own context)
        if depth > 0 and not lanes and len(set(_devices())) == 1:
            pre = engine.Prefetch(order, build, depth, discard)
            feed = iter(pre)

        def finish(ctx, gi, targets, cands, probes, ncand, nuniq, ids, nrows, lock=None):
            """An item's result into out / timings (lock: several lanes finish groups at once)."""
            members = items[gi]
            target_genomes = genomes_of(gi)
            ids_arr = np.asarray(ids, dtype=np.int64)
            if return_ids:
                res = ids
            else:
                seqs = [s for g in target_genomes for s in g.seqs]
                pos = cands.positions(ids_arr)
                which = np.searchsorted(targets.seq_off, pos, side="right") - 1
                local = pos - targets.seq_off[which]
                res = [seqs[q][o:o + probe_length]
                       for q, o in zip(which.tolist(), local.tolist())]
            if len(members) > 1:
                # a union: every group's own picks, in its own order; ids count from the group's first candidate
                cgrp = cands.groups()
                per_group = np.bincount(cgrp, minlength=len(members))
                first = np.concatenate([[0], np.cumsum(per_group)])
                pick_grp = cgrp[ids_arr] if ids_arr.size else np.zeros(0, dtype=np.int64)
                sel = np.argsort(pick_grp, kind="stable")          # (every group's picks together, in their order)
                g_sorted = pick_grp[sel]
                bounds = np.searchsorted(g_sorted, np.arange(len(members) + 1)).tolist()
                local = (ids_arr[sel] - first[g_sorted]).tolist() if return_ids else [res[q] for q in sel.tolist()]
                parts, units = [], 0.0
                for m, member in enumerate(members):
                    parts.append(local[bounds[m]:bounds[m + 1]])
                    units += float(per_group[m]) * float(sum(g.size() for g in target_genomes_grouped[member]))
            else:
                parts = [res]
                units = float(cands.n) * float(sum(g.size() for g in target_genomes))
            if lock is not None:
                lock.acquire()
            try:
                for member, part in zip(members, parts):
                    out[member] = part
                timings["candidates"] += ncand
                timings["unique_candidates"] += nuniq
                timings["probe_bp_units"] = timings.get("probe_bp_units", 0.0) + units
                _accumulate(timings, ctx, nrows, len(ids))
            finally:
                if lock is not None:
                    lock.release()

        try:
            if lanes:
                import threading
                nl = max(1, min(lane_count, len(order)))
                ctxs = _contexts(nl)
                # What an item costs a lane: its bases plus a fixed share (round 5: a 25-Mbase group takes 7-9 ms, the
                # 265-Mbase one 62 -- ~12 Mbases' worth of launches and read-backs per instance; with plain sizes the
                # lane that got the eight medium groups of S4 ran 40 ms longer than the lane with the large one)
                fixed = float(os.environ.get("CATCHHIP_LANE_FIXED_MBASES", "12")) * 1e6
                cost = {gi: sizes[gi] + fixed for gi in order}
                lists = [[order[j] for j in lane] for lane in _lpt([cost[gi] for gi in order], nl)]
                lane_of = {gi: li for li, lst in enumerate(lists) for gi in lst}
                if keep_input_order:
                    production = list(order)
                else:
                    # the order in which the lanes will ask for their groups if time goes as cost
                    # (round 6, measured and dropped: the lanes' first items produced smallest first, so that the device has
                    # work after a few milliseconds instead of the ~21 ms the largest group takes to pack: 0.1206 vs 0.1220 s)
                    production, at, clock = [], [0] * nl, [0.0] * nl
                    while len(production) < len(order):
                        li = min((l for l in range(nl) if at[l] < len(lists[l])), key=lambda l: (clock[l], l))
                        gi = lists[li][at[li]]
                        production.append(gi)
                        clock[li] += cost[gi]
                        at[li] += 1
                if keep_input_order:
                    # (the lanes then take their groups in production order too)
                    lists = [[gi for gi in production if lane_of[gi] == li] for li in range(nl)]
                cv = threading.Condition()
                built, errors = {}, []
                slots = threading.Semaphore(max(2, depth) * nl)
                res_lock = threading.Lock()

                # Builders: one (CATCHHIP_BUILDERS, a test hook, starts more when no random numbers are drawn while
                # building; each takes the next item of the production order when it is free).  Measured in round 5 on S4
                # from host strings: two builders, or lanes balanced by cost instead of size, leave the pass at 0.128-0.133 s
                # -- the three lanes' kernels share one device, and the pass is within ~10 % of the device work it holds
                # (96 ms of scans and solves one chain at a time + ~20 ms of front-end kernels + 0.5 GB of uploads).
                # (keep_input_order is set whenever building draws random numbers -- random anchors, a near-duplicate
                # filter --: ONE builder then, whatever the hook says, so that the draws stay in production order)
                nbuilders = 1 if keep_input_order else max(1, int(_lib.test_env("CATCHHIP_BUILDERS", "1")))
                next_item = [0]
                take_lock = threading.Lock()

                def producer(worker=0):
                    while True:
                        slots.acquire()
                        with take_lock:
                            at_ = next_item[0]
                            next_item[0] += 1
                        if at_ >= len(production) or errors:
                            slots.release()
                            return
                        gi = production[at_]
                        try:
                            # built on the upload context of the DEVICE whose lane will consume it: with CATCHHIP_DEVICES
                            # the lanes sit on several GPUs, and an object cannot change hands across devices
                            res = build(gi, engine.upload_context(ctxs[lane_of[gi]].device, index=worker))
                        except BaseException as exc:
                            with cv:
                                errors.append(exc)
                                cv.notify_all()
                            return
                        with cv:
                            built[gi] = res
                            cv.notify_all()

                def lane(li):
                    ctx = ctxs[li]
                    for gi in lists[li]:
                        with cv:
                            while gi not in built and not errors:
                                cv.wait()
                            if errors:
                                return
                            targets, cands, probes, ncand, nuniq = built.pop(gi)
                        slots.release()
                        try:
                            t_s0 = _time.perf_counter()
                            for h in (targets, cands, probes):
                                h.rebind(ctx)
                            if cands.n == 0:
                                logger.warning("There are no candidate probes for a "
                                               "grouping of genomes")
                            ids, nrows = engine.setcover_filter(
                                ctx, probes, targets, self.mismatches, self.lcf_thres,
                                self.island_of_exact_match, self.cover_extension, cands.n, None,
                                universe_p_of(gi), self.scan_mode)
                            t_s1 = _time.perf_counter()
                            finish(ctx, gi, targets, cands, probes, ncand, nuniq, ids, nrows, res_lock)
                            events.append(("solve lane %d" % li, gi, t_s0 - t_call, t_s1 - t_call))
                            events.append(("finish lane %d" % li, gi, t_s1 - t_call, _time.perf_counter() - t_call))
                        except BaseException as exc:
                            with cv:
                                errors.append(exc)
                                cv.notify_all()
                            slots.release()          # (the producer may be waiting for a slot)
                            try:
                                ctx.sync()           # kernels of this lane may still read the objects closed below
                            except Exception:        # noqa: BLE001 -- the first error is the one to report
                                pass
                        finally:
                            for h in (probes, cands, targets):
                                h.close()
                        if errors:
                            return

                prods = [threading.Thread(target=producer, args=(w,), name="catchhip-prefetch") for w in range(nbuilders)]
                threads = [threading.Thread(target=lane, args=(li,), name="catchhip-lane") for li in range(1, nl)]
                for t in prods:
                    t.start()
                for t in threads:
                    t.start()
                lane(0)
                for t in threads:
                    t.join()
                for _ in range(len(production) + nbuilders + 1):
                    slots.release()                  # let the builders run off the end of the list (or into the error flag)
                for t in prods:
                    t.join()
                for res in built.values():           # built, never consumed (after an error)
                    discard(res)
                if errors:
                    raise errors[0]
            else:
                for chunk in chunks:
                    ctxs = _contexts(len(chunk))
                    specs, held, built = [], [], []
                    try:
                        for ctx, gi in zip(ctxs, chunk):
                            if feed is not None:
                                got, res = next(feed)
                                assert got == gi
                                held.extend(res[:3])
                                targets, cands, probes, ncand, nuniq = res
                                for h in (targets, cands, probes):
                                    h.rebind(ctx)
                            else:
                                targets, cands, probes, ncand, nuniq = build(gi, ctx)
                                held.extend((targets, cands, probes))
                            built.append((targets, cands, probes, ncand, nuniq))
                            if cands.n == 0:
                                logger.warning("There are no candidate probes for a "
                                               "grouping of genomes")
                            specs.append((ctx, probes, targets, cands.n, None, universe_p_of(gi)))
                        results = engine.setcover_filter_many(
                            specs, self.mismatches, self.lcf_thres,
                            self.island_of_exact_match, self.cover_extension,
                            self.scan_mode)
                        for ctx, gi, b, (ids, nrows) in zip(ctxs, chunk, built, results):
                            finish(ctx, gi, b[0], b[1], b[2], b[3], b[4], ids, nrows)
                    finally:
                        for h in reversed(held):
                            h.close()
        finally:
            if pre is not None:
                pre.close()
        timings["pipe_events"] = events
        self.last_timings = timings
        return out

    def _filter_genomes_device_union(self, target_genomes_grouped, probe_length,
                                     probe_stride, seq_length_to_skip=None,
                                     near_duplicate_filter=None,
                                     max_bases=300_000_000):
        """_filter_genomes_device for many small groups (the clusters of a
        clustered design): the groups of a chunk share one targets / candidates
        / probes triple that carries group numbers -- duplicates are removed
        inside a group only, the MinHash filter (if any) runs over all groups in
        one pass with each group's own hash functions, the scan pairs a probe
        with its own group's genomes only, and one greedy solve over the
        disjoint union makes every group's own picks in its own order.
        max_bases: bases per instance (a probes object holds < 2^31 bytes of
        probe text: ~21 M candidates of 100 bases, one per 50 target bases; and
        the seed work list of a first scan is sized for 6 seeds per base, 68 B
        each until the rows are built: 300 Mbases -> ~120 GB at most)."""
        assert not self.identify and not self.avoided_genomes
        ngroups = len(target_genomes_grouped)
        out = [[] for _ in range(ngroups)]
        timings = dict(scan_ms=0.0, rows_ms=0.0, greedy_ms=0.0, picks=0,
                       rows=0, scan_launches=0, greedy_launches=0,
                       candidates=0, unique_candidates=0)
        ctx = engine.default_context()
        max_bases = int(float(_lib.test_env("CATCHHIP_UNION_MAX_MBASES", str(max_bases / 1e6))) * 1e6)     # (test hook)
        # the clusters of a clustered design may come as views of the genomes' storage (probe_designer.
        # ClusteredFragments: every member a single-sequence genome): sizes, targets and the selected probes' text
        # are then taken from the table, and no Genome / str is made per fragment
        views = target_genomes_grouped if hasattr(target_genomes_grouped, "table") else None
        group_bases = (views.group_bases().tolist() if views is not None
                       else [sum(g.size() for g in grp) for grp in target_genomes_grouped])
        chunks, at = [], 0
        while at < ngroups:
            chunk, bases = [], 0
            while at < ngroups:
                b = group_bases[at]
                if chunk and bases + b > max_bases:
                    break
                chunk.append(at)
                bases += b
                at += 1
            chunks.append(chunk)

        # Two stages, one chunk apart (CATCHHIP_PREFETCH_DEPTH, 0 = one after the other): a helper thread packs the
        # NEXT chunk's targets, enumerates and de-duplicates its candidates and runs the near-duplicate filter on the
        # upload context (the MinHash filter is ~45 dependent rounds per call: the GPU is mostly waiting), while this
        # thread draws the current chunk's anchors (NumPy, seconds at this scale), scans and solves on the compute
        # context.  The filter's hash functions / sampled positions come from `random`, the anchors from `np.random`:
        # each generator is only ever used by one of the two threads, in chunk order -- the same draws as one after
        # the other.
        depth = int(os.environ.get("CATCHHIP_PREFETCH_DEPTH", "2"))

        import time as _time
        stage_s = dict(pack_s=0.0, candidates_s=0.0, near_duplicates_s=0.0, anchors_s=0.0, solve_s=0.0,
                       ndf_ms=0.0, ndf_probes=0, ndf_pairs=0, ndf_kept=0)
        events = []      # (stage, chunk number, start, end) in seconds since the call began: the pipeline's timeline
        t_call = _time.perf_counter()

        # The front end runs on `workers` threads, each with its own stream (two; test hook CATCHHIP_FRONT_END_WORKERS; the
        # MinHash filter is ~45 dependent rounds of mostly small launches per chunk, 5.3 of the 6.8 s of
        # S5 x 1.0's filters: two of them side by side overlap).  The filter's draws from `random` are made for
        # every chunk first, in chunk order -- the stream of draws one chunk after the other would make.
        workers = max(1, int(_lib.test_env("CATCHHIP_FRONT_END_WORKERS", "2")))
        piped = depth > 0 and len(chunks) > 1
        drawn_ndf = {}
        if near_duplicate_filter is not None and not hasattr(near_duplicate_filter, "_draw_for_groups"):
            # (a filter that cannot draw ahead would draw from the global `random` on two threads at
            # once -- a nondeterministic selection; one builder keeps its draws in chunk order)
            workers = 1
        import threading as _threading
        stage_lock = _threading.Lock()       # (the builders add their stage times and counters side by side)
        if piped and near_duplicate_filter is not None and hasattr(near_duplicate_filter, "_draw_for_groups"):
            for ci, chunk in enumerate(chunks):
                drawn_ndf[ci] = near_duplicate_filter._draw_for_groups(len(chunk))
        chunk_no = {id(c): ci for ci, c in enumerate(chunks)}

        def build(chunk, worker=None):
            bctx = ctx if worker is None else engine.upload_context(index=worker)
            t0 = _time.perf_counter()
            if views is not None:
                genomes = views.table.take(np.concatenate([views.clusters[gi] for gi in chunk]))
                ngen = [len(views.clusters[gi]) for gi in chunk]
            else:
                genomes = [g.seqs for gi in chunk for g in target_genomes_grouped[gi]]
                ngen = [len(target_genomes_grouped[gi]) for gi in chunk]
            targets = engine.Targets(bctx, genomes)
            cands = None
            try:
                targets.set_groups(np.repeat(np.arange(len(chunk)), ngen))
                t1 = _time.perf_counter()
                cands = engine.Candidates(bctx, targets, probe_length, probe_stride, seq_length_to_skip)
                ncand, nuniq = cands.ncandidates, cands.n
                t2 = _time.perf_counter()
                if near_duplicate_filter is not None:
                    if chunk_no[id(chunk)] in drawn_ndf:
                        near_duplicate_filter._apply_to_grouped_candidates(cands, len(chunk), drawn_ndf[chunk_no[id(chunk)]])
                    else:
                        near_duplicate_filter._apply_to_grouped_candidates(cands, len(chunk))
                    # device time of the filter (HIP events around its launches on this worker's stream) and its work
                    ndf_ms_, cnt = bctx.kernel_ms(engine.PHASE_NDF)[0], bctx.ndf_counters()
                    with stage_lock:
                        stage_s["ndf_ms"] += ndf_ms_
                        stage_s["ndf_probes"] += cnt["probes"]
                        stage_s["ndf_pairs"] += cnt["pairs_compared"]
                        stage_s["ndf_tables"] = cnt["tables"]
                        stage_s["ndf_kept"] += cands.n
                bctx.sync()
                if bctx is not ctx:
                    # handed over here, while this stream is idle: a rebind later would wait for the NEXT chunk's
                    # filter rounds, which this thread queues on the same stream (1.1 s of S5 x 1.0)
                    for h in (targets, cands):
                        h.rebind(ctx)
                t3 = _time.perf_counter()
                with stage_lock:
                    stage_s["pack_s"] += t1 - t0
                    stage_s["candidates_s"] += t2 - t1
                    stage_s["near_duplicates_s"] += t3 - t2
                    events.append(("front", chunk_no[id(chunk)], t0 - t_call, t3 - t_call))
            except BaseException:
                for h in (cands, targets):
                    if h is not None:
                        h.close()
                raise
            return targets, cands, ncand, nuniq

        def discard(res):
            for h in (res[1], res[0]):
                h.close()

        def anchors(n):
            t0 = _time.perf_counter()
            got = _anchors_for_candidates(n, probe_length, self.mismatches, self.lcf_thres, self.kmer_probe_map_k)
            stage_s["anchors_s"] += _time.perf_counter() - t0
            events.append(("anchors", len(events), t0 - t_call, _time.perf_counter() - t_call))
            return got

        # Three stages since round 4 (depth >= 2): front end of chunk i + 2 | anchors of chunk i + 1 | scan and solve
        # of chunk i.  The anchors are NumPy draws and array work (2 s of S5 x 1.0), which release the interpreter
        # lock; np.random is still used by one thread only, in chunk order.
        pre = engine.PrefetchPool(chunks, build, workers, 1, discard) if piped else None
        pre2 = None
        if pre is not None and depth >= 2:
            first = iter(pre)

            def draw(_chunk):
                _c, res = next(first)
                try:
                    return res + (anchors(res[1].n),)
                except BaseException:
                    discard(res)
                    raise
            pre2 = engine.Prefetch(chunks, draw, 1, discard)
        if pre2 is not None:
            feed = iter(pre2)
        elif pre is not None:
            feed = ((c, res + (None,)) for c, res in pre)
        else:
            feed = ((c, build(c) + (None,)) for c in chunks)
        try:
            for chunk, (targets, cands, ncand, nuniq, drawn) in feed:
                logger.info("Groups %d..%d of %d as one instance", chunk[0] + 1, chunk[-1] + 1, ngroups)
                probes = None
                nrows, ids = 0, np.zeros(0, dtype=np.int64)
                try:
                    if views is not None:
                        vt = views.table.take(np.concatenate([views.clusters[gi] for gi in chunk]))
                        if self.coverage <= 1.0:
                            universe_p = [self.coverage] * len(vt)
                        else:      # (:761-792: a number of bases per genome)
                            universe_p = [float(min(self.coverage, n)) / n for n in vt.length.tolist()]
                    else:
                        seqs = [s for gi in chunk for g in target_genomes_grouped[gi] for s in g.seqs]
                        universe_p = [p for gi in chunk
                                      for p in self._make_universe_p(target_genomes_grouped[gi])]
                    timings["candidates"] += ncand
                    timings["unique_candidates"] += nuniq
                    anch = drawn if drawn is not None else anchors(cands.n)
                    t0 = _time.perf_counter()
                    probes = _probes_of_candidates(cands, anch)
                    ids, nrows = engine.setcover_filter(
                        ctx, probes, targets, self.mismatches, self.lcf_thres,
                        self.island_of_exact_match, self.cover_extension, cands.n,
                        None, universe_p, self.scan_mode, as_array=True)
                    stage_s["solve_s"] += _time.perf_counter() - t0
                    events.append(("solve", chunk_no[id(chunk)], t0 - t_call, _time.perf_counter() - t_call))
                    # candidate-probe x target-bp of the chunk: every cluster's own candidates x its bases
                    per_group = np.bincount(cands.groups(), minlength=len(chunk)) if cands.n else np.zeros(len(chunk), np.int64)
                    gbases = np.array([group_bases[gi] for gi in chunk], dtype=np.float64)
                    timings["probe_bp_units"] = timings.get("probe_bp_units", 0.0) + float(np.dot(per_group[:len(chunk)], gbases))
                    if ids.size:
                        grp = cands.groups()[ids]
                        pos = cands.positions(ids)
                        which = np.searchsorted(targets.seq_off, pos, side="right") - 1
                        local = pos - targets.seq_off[which]
                        if views is not None:
                            for g, q, o in zip(grp.tolist(), which.tolist(), local.tolist()):
                                out[chunk[g]].append(vt.string(q, o, o + probe_length))
                        else:
                            for g, q, o in zip(grp.tolist(), which.tolist(), local.tolist()):
                                out[chunk[g]].append(seqs[q][o:o + probe_length])
                finally:
                    if pre is not None:
                        ctx.sync()          # (objects built on the upload context go back to its cache: nothing may still read them)
                    for h in (probes, cands, targets):
                        if h is not None:
                            h.close()
                _accumulate(timings, ctx, nrows, int(ids.size))
                events.append(("out", chunk_no[id(chunk)], events[-1][3] if events else 0.0, _time.perf_counter() - t_call))
        finally:
            for p in (pre2, pre):
                if p is not None:
                    p.close()
        timings.update(stage_s)
        timings["union_chunks"] = len(chunks)
        timings["pipe_events"] = events
        self.last_timings = timings
        return out

    def _filter_strs_union(self, input_strs, target_genomes_grouped, todo,
                           selected, timings, max_bases=1 << 30,
                           max_candidates=(1 << 24) - 2):
        """Many independent groups (the clusters of a clustered design) as ONE
        instance per chunk: the groups' candidates and genomes share a probes /
        targets pair with group numbers (catchhip_*_set_groups), so the scan
        only pairs a probe with its own group's genomes; the union of disjoint
        set cover instances solved greedily makes, within every group, that
        group's own picks in its own order (gains never cross groups, ties go to
        the lowest id), and the solver's rounds cover all groups at once.
        Requires equal-length candidates everywhere (then every group would
        choose the same anchor rule and k as the union does, and the union's
        np.random draws are the groups' draws back to back) and no ranks."""
        ctx = engine.default_context()
        at = 0
        while at < len(todo):
            chunk, bases, ncand = [], 0, 0
            while at < len(todo):
                gi = todo[at]
                b = sum(g.size() for g in target_genomes_grouped[gi])
                if chunk and (bases + b > max_bases or
                              ncand + len(input_strs[gi]) > max_candidates):
                    break
                chunk.append(gi)
                bases += b
                ncand += len(input_strs[gi])
                at += 1
            logger.info("Set cover over groups %d..%d of %d as one instance",
                        chunk[0] + 1, chunk[-1] + 1, len(input_strs))
            counts = np.array([len(input_strs[gi]) for gi in chunk], dtype=np.int64)
            offsets = np.concatenate(([0], np.cumsum(counts)))
            strs = [s for gi in chunk for s in input_strs[gi]]
            k, uniq, owner, ep, eo = probe.anchor_table(
                strs, self.mismatches, self.lcf_thres,
                min_k=self.kmer_probe_map_k, k=self.kmer_probe_map_k,
                assume_unique=True)
            genomes = [g.seqs for gi in chunk for g in target_genomes_grouped[gi]]
            ngen = [len(target_genomes_grouped[gi]) for gi in chunk]
            universe_p = [p for gi in chunk
                          for p in self._make_universe_p(target_genomes_grouped[gi])]
            targets = engine.Targets(ctx, genomes)
            try:
                probes = engine.Probes(ctx, uniq, owner, ep, eo, k)
                try:
                    probes.set_groups(np.repeat(np.arange(len(chunk)), counts))
                    targets.set_groups(np.repeat(np.arange(len(chunk)), ngen))
                    ids, nrows = engine.setcover_filter(
                        ctx, probes, targets, self.mismatches, self.lcf_thres,
                        self.island_of_exact_match, self.cover_extension,
                        len(strs), None, universe_p, self.scan_mode)
                finally:
                    probes.close()
            finally:
                targets.close()
            ids = np.asarray(ids, dtype=np.int64)
            grp = np.searchsorted(offsets, ids, side="right") - 1
            for j, gi in enumerate(chunk):
                selected[gi] = (ids[grp == j] - offsets[j]).tolist()
            _accumulate(timings, ctx, nrows, len(ids))


def _lpt(costs, nbins):
    from catch_amd import parallel
    return parallel.lpt_assign(costs, nbins)


def _accumulate(timings, ctx, nrows, npicks):
    """Adds the device times (HIP events per phase) and work counters of the
    context's last fused filter call to `timings`."""
    for name, ph in (("scan_ms", engine.PHASE_SCAN), ("verify_ms", engine.PHASE_VERIFY),
                     ("rows_ms", engine.PHASE_ROWS), ("greedy_ms", engine.PHASE_GREEDY),
                     ("rounds_ms", engine.PHASE_GREEDY_ROUNDS), ("claim_ms", engine.PHASE_CLAIM)):
        ms, nl = ctx.kernel_ms(ph)
        timings[name] = timings.get(name, 0.0) + ms
        ln = name.replace("_ms", "_launches")
        timings[ln] = timings.get(ln, 0) + nl
    timings["rows"] = timings.get("rows", 0) + nrows
    timings["picks"] = timings.get("picks", 0) + npicks
    for k, v in ctx.counters().items():
        if k not in ("picks", "_"):
            timings[k] = timings.get(k, 0) + v


def _chunks_by_size(todo, size_of, width):
    """Groups in flight together.  Large groups (>= CATCHHIP_BIG_GROUP_BASES,
    default 8 Mbases) fill the GPU on their own and run one at a time, largest
    first (their kernels only get in each other's way: four S4 groups in flight
    took 498 ms per pass against 333 ms one after the other); the small ones
    run `width` at a time on their own streams."""
    import os
    big_bases = int(os.environ.get("CATCHHIP_BIG_GROUP_BASES", str(8_000_000)))
    big = [[gi] for gi in todo if size_of(gi) >= big_bases]
    small = [gi for gi in todo if size_of(gi) < big_bases]
    return big + [small[i:i + width] for i in range(0, len(small), width)]


_extra_ctxs = {}


def _devices():
    """Devices the groups in flight are spread over.  Default: the default
    context's device only (one process per GPU, as torchrun launches the
    bench).  CATCHHIP_DEVICES=all or a comma-separated list lets ONE process
    use several GPUs for independent groups (first row: whole
    groups to GPUs, no collective): every context has its own device, stream
    and allocator partition, and catchhip_setcover_filter_many already runs
    one host thread per context."""
    import os
    first = engine.default_context().device
    spec = os.environ.get("CATCHHIP_DEVICES", "")
    if not spec:
        return [first]
    if spec == "all":
        devs = list(range(engine.device_count()))
    else:
        devs = [int(x) for x in spec.split(",") if x.strip() != ""]
    devs = [d for d in devs if 0 <= d < engine.device_count()]
    if first in devs:                      # the default context keeps slot 0
        devs.remove(first)
    return [first] + devs


def _contexts(n):
    """The default context plus cached extra ones, one per group in flight,
    round-robin over _devices()."""
    first = engine.default_context()
    devs = _devices()
    out = [first]
    for i in range(1, n):
        dev = devs[i % len(devs)]
        key = (dev, i // len(devs))
        if key not in _extra_ctxs:
            _extra_ctxs[key] = engine.Context(dev)
        out.append(_extra_ctxs[key])
    return out


def _anchors_for_candidates(n, probe_length, mismatches, lcf_thres, map_k):
    """(k, "table" | "draws" | "entries", payload) for the device front end's n
    equal-length candidates: the pigeonhole table (payload None), random anchors
    as their np.random draws (the device sorts and de-duplicates them), or -- for
    probes longer than 256 + k -- as host-made entries.  The same np.random draws
    in all three forms."""
    if probe_length - map_k + 1 <= 256:
        k, draws = probe.anchor_draws_equal_length(n, probe_length, mismatches, lcf_thres,
                                                   min_k=map_k, k=map_k)
        return (k, "table", None) if draws is None else (k, "draws", draws)
    k, ep, eo = probe.anchor_entries_equal_length(n, probe_length, mismatches, lcf_thres,
                                                  min_k=map_k, k=map_k)
    return (k, "table", None) if ep is None else (k, "entries", (ep, eo))


def _probes_of_candidates(cands, anchors):
    k, kind, payload = anchors
    if kind == "draws":
        return cands.probes_from_draws(k, payload)
    if kind == "entries":
        return cands.probes(k, payload[0], payload[1])
    return cands.probes(k)
