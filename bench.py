#!/usr/bin/env python3
"""Benchmark of the SetCoverFilter hot path (K1 coverage scan + row build +
K2 greedy set cover) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload S4|S2|S3]

Workload (default S4 = BASELINE.json configs[3], the largest configuration
that fits one GPU; configs[4], the V-All-scale input, is `--workload S5`): the
seeded synthetic pool of 20 species / 20 groups (catch_amd/utils/synthetic.py,
SURVEY.md 8(d); the reference ships no input at this scale), `design.py
-pl 100 -ps 50 -m 2 -e 50 -c 1.0`.  `--workload S2` is configs[1] (the round-1
bench), `--workload S3` configs[2]'s genomes without the near-duplicate filter.

A step = one full pass of the hot path over every group with the packed
inputs already resident in HBM (targets as bit planes, de-duplicated
candidates as packed probe images + anchors): per group one fused
catchhip_setcover_filter call = hash-seeded coverage scan, bucketed row build,
frontier set-cover solver, selected ids back on the host.  Groups are
independent instances: the large ones run one after the other, largest first
(each fills the GPU), the small ones `--groups-in-flight` (default 4) at a
time on their own HIP streams.

N > 1 (launched by torch.distributed.run, one rank per GPU): ONE dataset,
strong scaling (catch_amd/parallel.py): a group above an even share of the
work is sharded over all ranks by universes -- every rank scans all candidates
against its range of genomes and the frontier solver exchanges one SUM
all-reduce of the per-candidate gains and one MAX all-reduce of the lost flags
per round over RCCL; the other groups go whole to ranks, longest first, with
no data-path collective.  The picks must equal the committed digests whatever N.

`value` is the resident-input rate the measurement contract prescribes (inputs
already in HBM when the timed region starts).  The wall-clock the metric also
names -- SURVEY 8(d) M2: pack + H2D + front end + scan + solve + ids out, from
host strings, nothing resident -- is measured in the same run through the
plugin's own path (SetCoverFilter._filter_genomes_device, which packs and
uploads group i + 1 on an upload context while group i is scanned and solved):
`m2_setcoverfilter_wall_s`, `value_incl_h2d`, and the same without the overlap
(`m2_serial_wall_s`).

Prints one JSON line on rank 0 (fields: README / DESIGN.md section 6).
"""
import argparse
import concurrent.futures
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# (the bench compares code paths through the library's test hooks -- the other solver family on S5, the E_dirty
# statistics of the row-parallel solver; they are only honoured with this set: csrc/internal.h, chip_test_env)
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")

import numpy as np  # noqa: E402

from catch_amd import engine, parallel, probe  # noqa: E402
from catch_amd.utils import synthetic  # noqa: E402

PROBE_LEN, STRIDE, MISMATCHES, EXT = 100, 50, 2, 50
SCAN_MODE = int(os.environ.get("CATCHHIP_SCAN_MODE", "0"))   # 0 auto, 1 general, 2 fast
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec
CONFIG_OF = {"S1": 0, "S2": 1, "S3": 2, "S4": 3, "S4i": 3, "S2i": 1, "S5": 4, "S5m": 4}


class ResidentGroup:
    """One group's packed inputs in HBM: targets (bytes + bit planes), unique
    candidates (device front end = candidate windows + DuplicateFilter) and the
    probes object (packed images + pigeonhole anchors)."""

    def __init__(self, ctx, index, genomes):
        self.ctx, self.index = ctx, index
        self.targets = engine.Targets(ctx, genomes)
        self.cands = engine.Candidates(ctx, self.targets, PROBE_LEN, STRIDE)
        k, ep, eo = probe.anchor_entries_equal_length(
            self.cands.n, PROBE_LEN, MISMATCHES, PROBE_LEN)
        self.probes = self.cands.probes(k, ep, eo)
        self.n_sets = self.cands.n
        self.G = self.targets.total
        self.n_genomes = len(genomes)

    def run(self):
        return engine.setcover_filter(self.ctx, self.probes, self.targets, MISMATCHES, PROBE_LEN, 0, EXT,
                                      self.n_sets, mode=SCAN_MODE, as_array=True)

    def close(self):
        self.probes.close()
        self.cands.close()
        self.targets.close()


class ResidentUnion:
    """Several small groups as ONE instance: targets / candidates / probes that carry group numbers (duplicates are
    removed inside a group only, the scan pairs a probe with its own group's genomes only, and one solve over the
    disjoint union makes every group's own picks in its own order) -- what SetCoverFilter does for the clusters of
    a clustered design and, in its default path, for groups below CATCHHIP_UNION_SMALL_BELOW_MBASES.  A group of a
    few Mbases is a chain of ~60 launches of microseconds each; seventeen such groups one after the other are
    seventeen chains, their union is one."""

    def __init__(self, ctx, indices, groups):
        self.ctx, self.indices = ctx, list(indices)
        self.index = "union of groups %s" % self.indices
        genomes = [g for i in self.indices for g in groups[i]]
        self.targets = engine.Targets(ctx, genomes)
        self.targets.set_groups(np.repeat(np.arange(len(self.indices)), [len(groups[i]) for i in self.indices]))
        self.cands = engine.Candidates(ctx, self.targets, PROBE_LEN, STRIDE)
        k, ep, eo = probe.anchor_entries_equal_length(self.cands.n, PROBE_LEN, MISMATCHES, PROBE_LEN)
        self.probes = self.cands.probes(k, ep, eo)
        self.n_sets = self.cands.n
        self.cgrp = self.cands.groups()
        self.first = np.concatenate([[0], np.cumsum(np.bincount(self.cgrp, minlength=len(self.indices)))])

    def run(self):
        """-> ({group index: its pick ids, counted from the group's first candidate}, rows)"""
        ids, nrows = engine.setcover_filter(self.ctx, self.probes, self.targets, MISMATCHES, PROBE_LEN, 0, EXT,
                                            self.n_sets, mode=SCAN_MODE, as_array=True)
        return self.split(ids), nrows

    def split(self, ids):
        """The union's picks -> {group index: that group's picks, in their order, counted from its first candidate}
        (int64 arrays: the ~10^5 picks never become Python lists in the timed step)."""
        ids = np.asarray(ids, dtype=np.int64)
        grp = self.cgrp[ids]
        order = np.argsort(grp, kind="stable")            # (one stable sort instead of a mask per group)
        g_sorted = grp[order]
        local = ids[order] - self.first[g_sorted]
        bounds = np.searchsorted(g_sorted, np.arange(len(self.indices) + 1)).tolist()
        return {gi: local[bounds[m]:bounds[m + 1]] for m, gi in enumerate(self.indices)}

    def close(self):
        self.probes.close()
        self.cands.close()
        self.targets.close()


class NdfGroup(ResidentGroup):
    """configs[2] (`--filter-with-lsh-hamming 2`): the targets are resident; a
    step = device front end (candidate windows + exact de-duplication) ->
    Hamming near-duplicate filter (K3: hash of 20 sampled positions per table,
    radix sort, pair verification, resolution rounds) -> scan + solve over the
    candidates the filter keeps."""
    HAMMING = 2

    def __init__(self, ctx, index, genomes):
        self.ctx, self.index = ctx, index
        self.targets = engine.Targets(ctx, genomes)
        self.G = self.targets.total
        self.n_genomes = len(genomes)
        self.cands = self.probes = None
        self.ndf_stats = {}
        self.run()                       # sizes (n_sets) for the unit count

    def run(self):
        import random
        from catch_amd.filter.near_duplicate_filter import NearDuplicateFilterWithHammingDistance
        if self.probes is not None:
            self.probes.close()
            self.cands.close()
        t0 = time.perf_counter()
        self.cands = engine.Candidates(self.ctx, self.targets, PROBE_LEN, STRIDE)
        windows, unique = self.cands.ncandidates, self.cands.n
        front_ms = self.ctx.kernel_ms(engine.PHASE_NDF)[0]
        random.seed(7)                   # the seed the committed digests were made with
        ndf = NearDuplicateFilterWithHammingDistance(self.HAMMING, PROBE_LEN)
        t1 = time.perf_counter()
        ndf._apply_to_candidates(self.cands)
        t2 = time.perf_counter()
        ms, nl = self.ctx.kernel_ms(engine.PHASE_NDF)
        cn = self.ctx.ndf_counters()
        self.ndf_stats = dict(ndf_ms=ms, ndf_launches=nl, ndf_wall_ms=(t2 - t1) * 1e3,
                              front_end_ms=front_ms, front_end_wall_ms=(t1 - t0) * 1e3, windows=windows,
                              unique_windows=unique, ndf_probes=cn["probes"], ndf_tables=cn["tables"],
                              ndf_pairs=cn["pairs_compared"], ndf_edges=cn["edges"], ndf_kept=self.cands.n)
        k, ep, eo = probe.anchor_entries_equal_length(self.cands.n, PROBE_LEN, MISMATCHES, PROBE_LEN)
        self.probes = self.cands.probes(k, ep, eo)
        self.n_sets = self.cands.n
        return ResidentGroup.run(self)


class ShardedGroup:
    """One group sharded over all ranks by universes: every rank holds the
    group's candidates (they come from all of its genomes) and the bit planes
    of its own contiguous range of genomes; a step = local scan + row build +
    the frontier rounds with two RCCL all-reduces each."""

    def __init__(self, W, index, genomes):
        self.W, self.index = W, index
        ctx = self.ctx = W.comm_ctx
        full = engine.Targets(ctx, genomes)
        self.cands = engine.Candidates(ctx, full, PROBE_LEN, STRIDE)   # keeps `full` alive
        k, ep, eo = probe.anchor_entries_equal_length(
            self.cands.n, PROBE_LEN, MISMATCHES, PROBE_LEN)
        self.probes = self.cands.probes(k, ep, eo)
        self.n_sets = self.cands.n
        b = parallel.split_universes([sum(len(s) for s in g) for g in genomes], W.size)
        self.range = (b[W.rank], b[W.rank + 1])
        self.targets = engine.Targets(ctx, genomes[b[W.rank]:b[W.rank + 1]])
        self.G_local = self.targets.total
        self.solve_ms = 0.0

    def step(self, stats=None):
        t0 = time.perf_counter()
        rows = engine.Rows.scan(self.ctx, self.probes, self.targets, MISMATCHES,
                                PROBE_LEN, 0, EXT, SCAN_MODE)
        t1 = time.perf_counter()
        shard = engine.Shard(rows, self.n_sets)
        try:
            ids = parallel.sharded_solve([shard], self.W.exchange_for([shard]), self.W.native_for([shard]))
        finally:
            shard.close()
            nrows = rows.n
            rows.close()
        if stats is not None:
            c = self.ctx
            st = dict(rows=nrows, picks=len(ids) if self.W.rank == 0 else 0,
                      sharded_scan_wall_ms=(t1 - t0) * 1e3,
                      sharded_solve_wall_ms=(time.perf_counter() - t1) * 1e3)
            for name, ph in (("scan_ms", engine.PHASE_SCAN), ("verify_ms", engine.PHASE_VERIFY),
                             ("rows_ms", engine.PHASE_ROWS)):
                ms, nl = c.kernel_ms(ph)
                st[name] = ms
                st[name.replace("_ms", "_launches")] = nl
            cn = c.counters()
            st.update(raw_hits=cn["raw_hits"], seed_hits=cn["seed_hits"], greedy_iters=cn["greedy_iters"])
            stats.append(st)
        return ids

    def close(self):
        self.probes.close()
        self.cands.close()
        self.targets.close()


class Stepper:
    """Runs the groups of this rank.  Large groups (>= BIG_BASES bases) fill
    the GPU on their own and run one after the other, largest first, on the
    default context; the small ones run `width` at a time, each lane (worker
    thread + context = HIP stream) owning the groups lpt_assign gave it --
    kernels of large groups only get in each other's way (seed_lookup went
    from 0.6-8 ms to 12 ms per launch with four S4 groups in flight)."""
    BIG_BASES = int(os.environ.get("CATCHHIP_BENCH_BIG_BASES", str(8_000_000)))
    UNION_BELOW = int(float(os.environ.get("CATCHHIP_BENCH_UNION_BELOW_MBASES", "32")) * 1e6)
    ndf = False        # configs[2]: the Hamming near-duplicate filter is part of the step

    def __init__(self, device, groups, indices, width):
        sizes = {i: sum(len(s) for g in groups[i] for s in g) for i in indices}
        big = sorted((i for i in indices if sizes[i] >= self.BIG_BASES),
                     key=lambda i: (-sizes[i], i))
        small = [i for i in indices if sizes[i] < self.BIG_BASES]
        self.width = max(1, min(width, len(small))) if small else 1
        self.ctxs = [engine.default_context() if w == 0 else engine.Context(device)
                     for w in range(self.width)]
        Group = NdfGroup if self.ndf else ResidentGroup
        self.big = [Group(self.ctxs[0], i, groups[i]) for i in big]
        lanes = parallel.lpt_assign([sizes[i] for i in small], self.width)
        self.lanes = [[Group(self.ctxs[w], small[j], groups[small[j]])
                       for j in lane] for w, lane in enumerate(lanes)]
        # the groups below UNION_BELOW bases as one instance (ResidentUnion); their own resident objects stay
        # for the passes that go group by group (-c 0.9, the CPU sample, the overlapped-groups figure)
        mid = [g for g in self.big if sizes[g.index] < self.UNION_BELOW]
        members = sorted([g.index for g in mid] + small)
        self.union = None
        self.union_pool = None
        if not self.ndf and self.UNION_BELOW > 0 and len(members) >= 2:
            self.big_alone = [g for g in self.big if sizes[g.index] >= self.UNION_BELOW]
            # the union instance on a stream (and host thread) of its own beside the large groups, which run one
            # after the other: each chain has a dozen host read-backs (sizes of work lists, rounds), and the other
            # chain's kernels fill them (CATCHHIP_BENCH_UNION_BESIDE=0: after them, on the same stream)
            beside = bool(self.big_alone) and os.environ.get("CATCHHIP_BENCH_UNION_BESIDE", "1") != "0"
            self.union_ctx = engine.Context(device) if beside else self.ctxs[0]
            self.union = ResidentUnion(self.union_ctx, members, groups)
            if beside:
                self.union_ctx.sync()
                # (the large groups on two or three streams of their own as well: 88.6 -> 86.3 / 86.0 ms, not kept)
                self.union_pool = concurrent.futures.ThreadPoolExecutor(1)
        for c in self.ctxs:
            c.sync()
        self.pool = (concurrent.futures.ThreadPoolExecutor(self.width)
                     if self.width > 1 else None)
        self.resident = self.big + [g for lane in self.lanes for g in lane]

    def _run_lane(self, lane, stats):
        out = []
        for g in lane:
            ids, nrows = g.run()
            if stats is not None:
                self._collect(g, ids, nrows, stats)
            out.append((g.index, ids))
        return out

    def step(self, stats=None, beside=True):
        """One pass over every group of this rank -> {group index: pick ids}.
        beside=False: the union instance after the large groups instead of beside them (per-kernel times that
        no other stream's kernels stretch)."""
        if self.union is not None:
            if self.union_pool is not None and beside:
                fut = self.union_pool.submit(self.union.run)
                out = dict(self._run_lane(self.big_alone, stats))
                per_group, nrows = fut.result()
            else:
                out = dict(self._run_lane(self.big_alone, stats))
                per_group, nrows = self.union.run()
            out.update(per_group)
            if stats is not None:
                self._collect(self.union, range(sum(len(ids) for ids in per_group.values())), nrows, stats)
            return out
        out = dict(self._run_lane(self.big, stats))
        small = [g for lane in self.lanes for g in lane]
        if len(small) > 1 and len(small) <= self.width and not self.ndf:
            # one group per lane: the C side's persistent helper threads
            specs = [(g.ctx, g.probes, g.targets, g.n_sets, None, None)
                     for g in small]
            res = engine.setcover_filter_many(specs, MISMATCHES, PROBE_LEN, 0,
                                              EXT, SCAN_MODE)
            for g, (ids, nrows) in zip(small, res):
                out[g.index] = ids
                if stats is not None:
                    self._collect(g, ids, nrows, stats)
            return out
        if self.pool is None:
            parts = [self._run_lane(lane, stats) for lane in self.lanes]
        else:
            per = [[] if stats is not None else None for _ in self.lanes]
            futs = [self.pool.submit(self._run_lane, lane, st)
                    for lane, st in zip(self.lanes, per)]
            parts = [f.result() for f in futs]
            if stats is not None:
                for st in per:
                    stats.extend(st)
        out.update({i: ids for part in parts for i, ids in part})
        return out

    def _collect(self, g, ids, nrows, stats):
        c = g.ctx
        st = dict(rows=nrows, picks=len(ids))
        for name, ph in (("scan_ms", engine.PHASE_SCAN),
                         ("verify_ms", engine.PHASE_VERIFY),
                         ("vcount_ms", engine.PHASE_VCOUNT),
                         ("rows_ms", engine.PHASE_ROWS),
                         ("greedy_ms", engine.PHASE_GREEDY),
                         ("rounds_ms", engine.PHASE_GREEDY_ROUNDS),
                         ("claim_ms", engine.PHASE_CLAIM)):
            ms, nl = c.kernel_ms(ph)
            st[name] = ms
            st[name.replace("_ms", "_launches")] = nl
        st.update(c.counters())
        st.update(getattr(g, "ndf_stats", {}))
        # the claim kernel of the row-parallel solver streams every alive record (8 B since round 4) and looks at
        # the owner word (8 B) of every flagged word of it
        st["claim_bytes"] = (8.0 * st["flat_rows_streamed"] + 8.0 * st["flat_owner_words"]) if st["claim_launches"] else 0.0
        stats.append(st)

    def sync(self):
        for c in self.ctxs:
            c.sync()
        if getattr(self, "union_pool", None) is not None:
            self.union_ctx.sync()

    def close(self):
        for g in self.resident:
            g.close()
        if self.union is not None:
            self.union.close()
        if getattr(self, "union_pool", None) is not None:
            self.union_pool.shutdown()
            self.union_pool = None
        if self.pool is not None:
            self.pool.shutdown()


def _newest_profile(stem, workload, scale):
    """The newest committed record profiles/r<NN>_<stem>_<workload>.json whose `workload` / `scale` are this run's
    (rounds in descending order, so nothing has to be re-stamped by hand when a round commits new passes); the
    record gets `_file` and `_round`.  None when there is none: counters cannot be collected from inside the timed
    run, so what the line reports as PMC traffic is read back from the committed rocprofv3 passes of this command."""
    import glob
    import re
    found = []
    for path in glob.glob(os.path.join(REPO, "profiles", "r*_%s_%s.json" % (stem, workload))):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        if m:
            found.append((int(m.group(1)), path))
    for rnd, path in sorted(found, reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("workload") == workload and float(rec.get("scale", 1.0)) == float(scale):
            rec["_file"] = os.path.relpath(path, REPO)
            rec["_round"] = rnd
            return rec
    return None


def _pmc_record(workload, scale):
    """The committed rocprofv3 PMC passes of this same command (profiles/r<NN>_pmc_traffic_<workload>.json, made by
    tools/collect_profiles.sh: FETCH_SIZE and WRITE_SIZE in separate runs), newest round first."""
    return _newest_profile("pmc_traffic", workload, scale) if scale == 1.0 else None


def pmc_traffic(unit, workload, scale):
    """HBM-side bytes per launch of `unit` (a unit of the record, or one kernel by name): 2 x FETCH_SIZE +
    WRITE_SIZE -- FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes."""
    rec = _pmc_record(workload, scale)
    if rec is None:
        return None
    try:
        k = rec["units"].get(unit) or rec["kernels"][unit + "_kernel"]
        return (2.0 * k["FETCH_SIZE_KB_per_launch"] + k["WRITE_SIZE_KB_per_launch"]) * 1024.0
    except KeyError:
        return None


def traffic_per_step(workload, scale):
    """GB of counted traffic per step and unit (2 x FETCH_SIZE + WRITE_SIZE), from the same record."""
    rec = _pmc_record(workload, scale)
    if rec is None or "steps_in_run" not in rec:
        return None
    out = {"source": rec["_file"], "formula": "(2 x FETCH_SIZE + WRITE_SIZE) summed over the unit's launches / steps of the run"}
    tot = 0.0
    for name, u in rec["units"].items():
        gb = (2.0 * u["FETCH_SIZE_KB_per_launch"] + u["WRITE_SIZE_KB_per_launch"]) * 1024.0 * u["launches"] / rec["steps_in_run"] / 1e9
        out[name + "_GB"] = gb
        if name in rec.get("disjoint_units", []):
            tot += gb
    out["total_GB"] = tot
    return out


def summarize_checks(checks):
    """Verdicts of catchhip_rows_cover_check (csrc/check.hip: the picks replayed per universe in pick order by
    kernels that share nothing with the solvers) over the instances of one step."""
    if not checks:
        return None
    tot = {k: int(sum(c[k] for c in checks)) for k in ("picks", "rows", "universes", "picks_without_gain", "universes_short",
                                                         "bad_pick_ids", "universe_bases", "covered_bases")}
    tot["instances"] = len(checks)
    tot["ok"] = tot["picks_without_gain"] == 0 and tot["universes_short"] == 0 and tot["bad_pick_ids"] == 0
    tot["what"] = ("independent replay of every instance's picks in pick order: every pick covered a new position at its "
                   "turn, every universe is covered to |U| - int(|U| - p |U|)")
    return tot


def ndf_bucket_independence(kept, positions, dist_thres):
    """The Hamming near-duplicate filter's guarantee, checked with numpy: no two KEPT probes that share a bucket
    (equal characters at a table's sampled positions; catch/utils/lsh.py:289-320) are within dist_thres of each other.
    kept: uint8 array [n, L].  -> (pairs compared, violations)."""
    n = kept.shape[0]
    pairs = bad = 0
    for pos in positions:
        key = np.ascontiguousarray(kept[:, pos])
        v = key.view(np.dtype((np.void, key.shape[1]))).ravel()
        order = np.argsort(v, kind="stable")
        sv = v[order]
        head = np.ones(n, dtype=bool)
        head[1:] = sv[1:] != sv[:-1]
        run = np.cumsum(head) - 1
        size = np.bincount(run)
        mx = int(size.max()) if n else 0
        for d in range(1, mx):
            i = np.nonzero(run[d:] == run[:-d])[0]
            if i.size == 0:
                break
            a, b = kept[order[i]], kept[order[i + d]]
            dist = (a != b).sum(axis=1)
            pairs += int(i.size)
            bad += int((dist <= dist_thres).sum())
    return pairs, bad


def digest(ids):
    a = np.sort(np.asarray(ids, dtype=np.int64))
    return hashlib.sha256(a.astype("<i8").tobytes()).hexdigest()


def digest_in_order(ids):
    """Order-sensitive: the solvers return the picks in the sequential pick order."""
    return hashlib.sha256(np.asarray(ids, dtype="<i8").tobytes()).hexdigest()


def golden_digests(workload, scale):
    """tests/golden/full_size_picks.json (made in the authoring container by
    tests/golden/make_full_size.py with the pinned CPU oracle)."""
    key = workload if scale == 1.0 else "%s:%g" % (workload, scale)
    try:
        with open(os.path.join(REPO, "tests", "golden", "full_size_picks.json")) as f:
            return {g["group"]: g for g in json.load(f)[key]["groups"]}
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(groups, sample, budget_s=20.0, mid=None):
    """The CPU oracle (oracle/, plain C) timed on a bounded sample of the
    workload's groups: per group the threaded per-sequence scans (all host
    cores, like the reference's process pool) and then the line-by-line greedy
    restatement (one core, like the reference's one process per instance).
    mid: one mid-size group (25-40 Mbases) on top, once, with the oracle's
    lazy-evaluation greedy (pinned to the line-by-line one; that one would take
    minutes there) -- so that the like-for-like ratio is not taken on the
    smallest groups alone.  Reported beside the GPU number, never part of it."""
    from catch_amd.filter import candidate_probes
    from oracle import oracle as orc
    orc.build()
    cores = orc.set_threads(orc.hw_threads())
    sel, units, cand_s = {}, 0, 0.0
    t0 = time.perf_counter()
    reps = 0
    while True:
        for gi in sample:
            genomes = groups[gi]
            tc = time.perf_counter()
            seqs = [s for g in genomes for s in g]
            cands = list(dict.fromkeys(
                candidate_probes.candidate_strings_from_sequences(
                    seqs, PROBE_LEN, STRIDE)))
            cand_s += time.perf_counter() - tc
            ids = orc.set_cover_filter([cands], [genomes], MISMATCHES, PROBE_LEN,
                                       coverage=1.0, cover_extension=EXT)[0]
            sel[gi] = ids
            units += len(cands) * sum(len(s) for s in seqs)
        reps += 1
        el = time.perf_counter() - t0 - cand_s
        if el >= budget_s / 2 or reps >= 50:
            break
    el = time.perf_counter() - t0 - cand_s
    mid_rec = None
    if mid is not None:
        genomes = groups[mid]
        seqs = [s for g in genomes for s in g]
        cands = list(dict.fromkeys(candidate_probes.candidate_strings_from_sequences(seqs, PROBE_LEN, STRIDE)))
        tm = time.perf_counter()
        ids = orc.set_cover_filter([cands], [genomes], MISMATCHES, PROBE_LEN, coverage=1.0, cover_extension=EXT,
                                   lazy=True)[0]
        mid_s = time.perf_counter() - tm
        sel[mid] = ids
        mid_rec = dict(group=mid, bases=sum(len(s) for s in seqs), candidates=len(cands), seconds=mid_s,
                       value=len(cands) * sum(len(s) for s in seqs) / mid_s, unit="probe*bp/s",
                       greedy="lazy evaluation over bitmaps (oracle.orc_lazy_greedy, pinned to the line-by-line "
                              "restatement; ~40 x faster than it)")
    orc.set_threads(1)
    return dict(value=units / el, unit="probe*bp/s", cores=cores, kind="port",
                seconds=el, passes=reps, mid_group=mid_rec,
                sample="groups %s of the bench workload (the smallest by bases, "
                       "%d of %d groups), %d pass%s through the plain-C oracle: "
                       "seed-and-extend scan on %d OpenMP threads, interval-set "
                       "greedy on 1 (as the reference: pool for the scans, one "
                       "process per instance for the solve)%s"
                       % (sample, len(sample), len(groups), reps,
                          "" if reps == 1 else "es", cores,
                          "" if mid_rec is None else "; plus group %d (%.1f Mbases) once, see mid_group" % (mid, mid_rec["bases"] / 1e6))), sel


def cpu_baseline_s3(ctx, scale=0.03):
    """configs[2] on the CPU: candidate windows -> Hamming near-duplicate
    filter (oracle's C port, one core: the reference's filter is sequential) ->
    set cover (threaded scans + greedy), on S3 x `scale`; the GPU runs the same
    chain on the same input and must select the same ids."""
    import random
    from catch_amd.filter import candidate_probes
    from catch_amd.filter.near_duplicate_filter import NearDuplicateFilterWithHammingDistance
    from oracle import oracle as orc
    orc.build()
    genomes = synthetic.dataset("S3", scale=scale)[0]
    seqs = [s for g in genomes for s in g]
    cores = orc.set_threads(orc.hw_threads())
    strs = candidate_probes.candidate_strings_from_sequences(seqs, PROBE_LEN, STRIDE)
    t0 = time.perf_counter()
    random.seed(7)
    pos = orc.lsh_draw_positions(orc.lsh_num_tables(NdfGroup.HAMMING, PROBE_LEN, 20), 20, PROBE_LEN)
    kept = orc.ndf_hamming_c(strs, NdfGroup.HAMMING, pos)
    exp = orc.set_cover_filter([kept], [genomes], MISMATCHES, PROBE_LEN, coverage=1.0,
                               cover_extension=EXT)[0]
    el = time.perf_counter() - t0
    orc.set_threads(1)
    g = NdfGroup(ctx, 0, genomes)
    ctx.sync()
    t1 = time.perf_counter()
    ids, _ = g.run()
    ctx.sync()
    gpu_s = time.perf_counter() - t1
    same = g.n_sets == len(kept) and sorted(ids) == sorted(exp)
    units = float(len(kept)) * sum(len(s) for s in seqs)
    g.close()
    return dict(value=units / el, unit="probe*bp/s", cores=cores, kind="port", seconds=el, passes=1,
                sample="S3 x %g (%d records, %d bp, %d windows -> %d after the Hamming filter): plain-C oracle, "
                       "near-duplicate filter on 1 core (sequential in the reference), scans on %d OpenMP "
                       "threads, greedy on 1" % (scale, len(genomes), sum(len(s) for s in seqs), len(strs),
                                                 len(kept), cores)), gpu_s, same


def cpu_baseline_s5(n_genomes=40):
    """configs[4]'s filters on the CPU: the oracle's chain (candidate windows -> MinHash near-duplicate filter 0.6 ->
    set cover with -m 5 -e 50 and 20 random anchors per probe) on the first `n_genomes` genomes of the S5 generator as
    ONE cluster -- bounded to tens of seconds: the oracle's MinHash filter is the reference's Python loop with the hash
    in C, ~1 ms per candidate.  The clustering pre-step is not in the sample (the oracle's needs minutes for a few
    thousand fragments).  The GPU runs the same chain on the same input (same `random` / `np.random` seeds) and must
    select the same probes."""
    import random
    from catch_amd import genome
    from catch_amd.filter import candidate_probes, near_duplicate_filter, set_cover_filter
    from oracle import oracle as orc
    orc.build()
    genomes = synthetic.dataset("S5", scale=0.002)[0][:n_genomes]
    seqs = [s for g in genomes for s in g]
    nbases = sum(len(s) for s in seqs)
    cores = orc.set_threads(orc.hw_threads())
    t0 = time.perf_counter()
    cands = candidate_probes.candidate_strings_from_sequences(seqs, PROBE_LEN, STRIDE)
    random.seed(21)
    np.random.seed(22)
    params = orc.minhash_draw_params(orc.minhash_num_tables(0.6), 3)
    kept = orc.ndf_minhash(cands, 0.6, params)
    t1 = time.perf_counter()
    exp = orc.set_cover_filter([kept], [genomes], 5, PROBE_LEN, coverage=1.0, cover_extension=EXT, lazy=True)[0]
    el = time.perf_counter() - t0
    orc.set_threads(1)
    want = sorted(kept[j] for j in exp)
    gobjs = [[genome.Genome.from_one_seq(g[0]) for g in genomes]]
    gpu_s, got = None, None
    for _ in range(2):          # (the second pass: pools and lazily built tables warm)
        random.seed(21)
        np.random.seed(22)
        ndf = near_duplicate_filter.NearDuplicateFilterWithMinHash(0.6)
        scf = set_cover_filter.SetCoverFilter(mismatches=5, lcf_thres=PROBE_LEN, coverage=1.0, cover_extension=EXT,
                                              kmer_probe_map_k=20)
        t2 = time.perf_counter()
        got = scf._filter_genomes_device_union(gobjs, PROBE_LEN, STRIDE, None, ndf)
        gpu_s = time.perf_counter() - t2
    same = sorted(got[0]) == want
    units = float(len(kept)) * nbases
    return dict(value=units / el, unit="probe*bp/s", cores=cores, kind="port", seconds=el, passes=1,
                ndf_seconds=t1 - t0, set_cover_seconds=el - (t1 - t0),
                sample="the first %d genomes of the S5 generator (%d bp) as one cluster: %d windows -> %d after the MinHash "
                       "filter -> %d probes; the oracle's MinHash filter (the reference's Python loop, hash in C, 1 core), "
                       "scans on %d OpenMP threads, lazy greedy on 1; clustering not in the sample"
                       % (len(genomes), nbases, len(cands), len(kept), len(exp), cores)), gpu_s, same


def m2_passes(groups, steps, warm, depth):
    """SURVEY 8(d) M2 through the plugin: every pass starts from the host's
    sequence strings (nothing resident) and ends with the selected ids on the
    host.  depth = groups the upload context may run ahead (0: no overlap)."""
    from catch_amd import genome
    from catch_amd.filter.set_cover_filter import SetCoverFilter
    from collections import OrderedDict
    gobjs = [[genome.Genome(list(g)) if len(g) == 1 else
              genome.Genome.from_chrs(OrderedDict((str(i), s) for i, s in enumerate(g)))
              for g in grp] for grp in groups]
    f = SetCoverFilter(mismatches=MISMATCHES, lcf_thres=PROBE_LEN, coverage=1.0,
                       cover_extension=EXT)
    f.scan_mode = SCAN_MODE
    old = os.environ.get("CATCHHIP_PREFETCH_DEPTH")
    os.environ["CATCHHIP_PREFETCH_DEPTH"] = str(depth)
    times, ids = [], None
    try:
        for i in range(warm + steps):
            t = time.perf_counter()
            ids = f._filter_genomes_device(gobjs, PROBE_LEN, STRIDE, return_ids=True)
            dt = time.perf_counter() - t
            if i >= warm:
                times.append(dt)
    finally:
        if old is None:
            os.environ.pop("CATCHHIP_PREFETCH_DEPTH", None)
        else:
            os.environ["CATCHHIP_PREFETCH_DEPTH"] = old
    return times, dict(enumerate(ids))


def digests_ok(gold, picks):
    if gold is None:
        return None
    return all(len(ids) == gold[gi]["n_picks"] and digest(ids) == gold[gi]["picks_sha256"]
               and ("picks_in_order_sha256" not in gold[gi]
                    or digest_in_order(ids) == gold[gi]["picks_in_order_sha256"])
               for gi, ids in picks.items() if gi in gold)


def seed_verify_bytes(seeds, dropped, hits, L=PROBE_LEN):
    """Algorithmic bytes of the seed-verify launches: per live seed a 12-B work item +
    0.375 B/base of the (L+32)-base target window and of the L-base probe + a 4-B rank, per dropped list
    entry 4 B, per hit a 16-B record."""
    live = seeds - dropped
    return live * (12 + 0.375 * (L + 32) + 0.375 * L + 4) + 4.0 * dropped + 16.0 * hits


VALU_PEAK_GWIPS = 1024 * 2.4 / 4.0     # wave-instructions / ns: 256 CUs x 4 SIMDs, a 64-wide VALU instruction occupies its SIMD for 4 cycles at 2.4 GHz


def units_record(workload, scale):
    """profiles/r<NN>_pmc_<workload>.json of the newest round that has one (tools/collect_units.sh): kernel time,
    FETCH_SIZE / WRITE_SIZE and SQ_INSTS_VALU of the committed rocprofv3 passes of this command, by unit of the hot
    path; None if there is none for this run."""
    return _newest_profile("pmc", workload, scale)


def valu_figures(unit, steps_in_run, device_ms_per_step):
    """VALU side of a unit that is neither HBM- nor MFMA-bound: wave-instructions per ns over the unit's device time
    against the issue peak, and the counters' own busy-cycle fraction."""
    if not unit or not unit.get("SQ_INSTS_VALU") or device_ms_per_step <= 0:
        return None
    per_step = unit["SQ_INSTS_VALU"] / max(steps_in_run, 1)
    ach = per_step / (device_ms_per_step * 1e6)
    return dict(bound="valu", achieved=ach, peak=VALU_PEAK_GWIPS, unit="wave-instructions/ns", frac=ach / VALU_PEAK_GWIPS,
                insts_per_step=per_step, issue_frac_of_busy_cycles=unit.get("valu_issue_frac"),
                source="%s (rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES, a run of its own)" % unit.get("_file", "profiles/r*_pmc_*.json"))


def closer_roof(roof, vf):
    """bound / achieved / peak / unit / frac of `roof` become those of the roof the unit is closer to (HBM as priced, or
    the VALU issue rate of the counters); the other one stays beside it."""
    hbm = {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
    out = dict(roof)
    if vf["frac"] > roof["frac"]:
        out.update({k: vf[k] for k in ("bound", "achieved", "peak", "unit", "frac")})
        out["hbm"] = hbm
    out["valu"] = vf
    out["bound_note"] = ("neither roof is near: the unit's passes are chains of dependent scattered loads (latency-bound); `bound` "
                         "names the roof it is closer to, the other roof's figures are beside it")
    return out


def s5_roofline(dom, dms, dbytes, dl, args):
    """The roofline object of the configs[4] leg for its dominant unit.  HBM: SURVEY 8(d)'s algorithmic bytes over the
    unit's device time (HIP events on the streams it runs on).  The counters of the committed rocprofv3 passes of this
    command (the newest profiles/r<NN>_pmc_S5.json, tools/collect_units.sh) give the traffic (2 x FETCH_SIZE + WRITE_SIZE per
    filter call) and, because the near-duplicate filter is neither HBM- nor MFMA-bound, the VALU issue fraction
    SQ_INSTS_VALU x 4 cycles / (256 CUs x 4 SIMDs x busy cycles) beside it."""
    def gbs(b, t_ms):
        return b / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
    roof = dict(bound="hbm", kernel=dom, achieved=gbs(dbytes, dms), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=gbs(dbytes, dms) / HBM_PEAK_GBS, traffic=None,
                algorithmic_bytes_per_launch=dbytes / max(dl, 1), avg_launch_ms=dms / max(dl, 1),
                launches_per_step=dl, device_ms_per_step=dms)
    rec = units_record(args.workload, args.scale)
    if rec is None:
        return roof
    unit = rec["units"].get("ndf" if dom.startswith("MinHash") else
                            "join_verify" if "verify" in dom else "rows_build" if "row build" in dom else "solver_round")
    if unit:
        per_step = 1.0 / max(rec.get("steps_in_run", 1), 1)
        tb = (2.0 * unit["FETCH_SIZE_KB"] + unit["WRITE_SIZE_KB"]) * 1024.0 * per_step
        roof["traffic"] = tb / max(dl, 1)
        roof["traffic_bytes_per_step"] = tb
        roof["traffic_source"] = "%s: (2 x FETCH_SIZE + WRITE_SIZE) of the unit's kernels / steps of the run / launches" % rec["_file"]
        roof["traffic_source_round"] = rec["_round"]
        unit = dict(unit, _file=rec["_file"])
        vf = valu_figures(unit, rec.get("steps_in_run", 1), dms)
        if vf is not None:
            roof = closer_roof(roof, vf)
    return roof


def bench_design_large(args):
    """--workload S5 = BASELINE configs[4]: the design_large chain on the
    synthetic V-All-shaped pool (one FASTA's worth of genomes, ONE group):
    50-kb fragments -> MinHash-signature clustering at 0.15 -> per cluster
    candidate windows, MinHash near-duplicate filter 0.6, set cover with
    -m 5 -e 50 (20 random anchors per probe) -- bin/design.py:502,583,753,794,846,
    catch/filter/probe_designer.py:78-184.  A step starts from the host's
    sequence strings (nothing resident: this IS the M2 figure) and ends with the
    selected probe strings on the host; Probe objects / FASTA output are not
    part of it.  One GPU."""
    import itertools
    import random
    from catch_amd import genome
    from catch_amd.filter import near_duplicate_filter, probe_designer, set_cover_filter
    t_gen = time.perf_counter()
    genomes = synthetic.dataset(args.workload, scale=args.scale)[0]     # S5, or S5m: its first 40 species
    gen_s = time.perf_counter() - t_gen
    gobjs = [[genome.Genome.from_one_seq(g[0]) for g in genomes]]
    bases = sum(len(s) for g in genomes for s in g)
    pool0 = engine.pool_stats()

    def one_step(env=None):
        old = {}
        for k, v in (env or {}).items():
            old[k] = os.environ.get(k)
            os.environ[k] = v
        try:
            random.seed(21)
            np.random.seed(22)
            ndf = near_duplicate_filter.NearDuplicateFilterWithMinHash(0.6)
            scf = set_cover_filter.SetCoverFilter(mismatches=5, lcf_thres=PROBE_LEN, coverage=1.0,
                                                  cover_extension=EXT, kmer_probe_map_k=20)
            pd = probe_designer.ProbeDesigner(gobjs, [ndf, scf], probe_length=PROBE_LEN, probe_stride=STRIDE,
                                              cluster_threshold=0.15, cluster_merge_after=scf,
                                              cluster_method="choose", cluster_fragment_length=50000)
            t0 = time.perf_counter()
            clusters = pd._cluster_genomes()
            t1 = time.perf_counter()
            mode = pd._device_front_end_mode(clusters, ndf, scf)
            run = scf._filter_genomes_device if mode == "per group" else scf._filter_genomes_device_union
            chosen = run(clusters, PROBE_LEN, STRIDE, None, ndf)
            t2 = time.perf_counter()
            probes = list(dict.fromkeys(itertools.chain(*chosen)))
            t3 = time.perf_counter()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        tm = dict(scf.last_timings)
        tm.update(cluster_s=t1 - t0, filter_s=t2 - t1, merge_s=t3 - t2, wall_s=t3 - t0,
                  clusters=len(clusters), mode=mode,
                  # (clusters that come as views of the genomes' storage are not iterated: that would build their Genomes)
                  fragments=(sum(len(c) for c in clusters.clusters) if hasattr(clusters, "clusters") else sum(len(c) for c in clusters)))
        for k, v in getattr(pd, "cluster_timings", {}).items():
            tm["cluster_" + k] = v
        return probes, tm

    def probes_digest(probes):
        return hashlib.sha256("\n".join(sorted(set(probes))).encode()).hexdigest()

    for _ in range(args.warmup):
        one_step()
    engine.default_context().sync()
    t0 = time.perf_counter()
    steps = []
    probes = None
    for _ in range(args.steps):
        probes, tm = one_step()
        steps.append(tm)
    engine.default_context().sync()
    elapsed = time.perf_counter() - t0
    K = args.steps
    per = {k: sum(st[k] for st in steps) / K for k in steps[0] if isinstance(steps[0][k], (int, float))}
    units = per.get("probe_bp_units", 0.0)
    dg = probes_digest(probes)
    gold = None
    try:
        with open(os.path.join(REPO, "tests", "golden", "full_size_picks.json")) as f:
            gold = json.load(f).get("%s:%g" % (args.workload, args.scale), {}).get("design")
    except (OSError, ValueError):
        pass
    seeds, dropped, hits = per.get("seed_hits", 0), per.get("seeds_dropped", 0), per.get("raw_hits", 0)
    vb = seed_verify_bytes(seeds, dropped, hits)
    rows = per.get("rows", 0)

    def gbs(b, t_ms):
        return b / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
    lookup_ms = per.get("scan_ms", 0.0) - per.get("verify_ms", 0.0)
    # SURVEY 8(d) K3 for the MinHash family: N probes x (L characters + T tables x (8 + 4) B x 4 radix passes, read and
    # written) + per compared pair the two k-mer code rows (4 B per distinct k-mer)
    N3, T3, C3 = per.get("ndf_probes", 0), per.get("ndf_tables", 0), per.get("ndf_pairs", 0)
    k3_bytes = N3 * (PROBE_LEN + T3 * 12.0 * 4 * 2) + C3 * 2.0 * 4 * (PROBE_LEN - 10 + 1)
    n_ndf_calls = max(1, int(round(per.get("union_chunks", 1))))
    cands_ms = {"MinHash near-duplicate filter (K3: mh_kmer + mh_keys_all + 25 radix sorts + ndf_lazy / ndf_probe passes)":
                (per.get("ndf_ms", 0.0), k3_bytes, n_ndf_calls),
                "seed_verify4_kernel (K1 verify)": (per.get("verify_ms", 0.0), vb, per.get("verify_launches", 0)),
                "bucketed row build": (per.get("rows_ms", 0.0), 44.0 * hits + 40.0 * rows, per.get("rows_launches", 0)),
                "frontier solver rounds (K2)": (per.get("rounds_ms", 0.0),
                                                12.0 * per.get("rows_recounted", 0) + 8.0 * per.get("bitmap_words_read", 0),
                                                per.get("rounds_launches", 0))}
    dom = max(cands_ms, key=lambda k: cands_ms[k][0])
    dms, dbytes, dl = cands_ms[dom]
    out = {
        "metric": "candidate-probe x target-bp / s through the design_large chain "
                  "(clustering + MinHash near-duplicate filter + SetCoverFilter), from host strings",
        "value": units * K / elapsed, "unit": "probe*bp/s", "n_gpus": 1, "steps": K, "warmup": args.warmup,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32 bit-planes / u64 bitmap (integer)", "data": "synthetic",
        "config": {"workload": args.workload + " (BASELINE configs[4]%s): design_large defaults on %d genomes, %d bp in one group: "
                               "-m 5 -e 50 -pl 100 -ps 50, cluster 0.15 from 50-kb fragments, MinHash NDF 0.6"
                               % ("" if args.scale == 1.0 else " scaled x%g" % args.scale, len(genomes), bases),
                   "scale": args.scale, "clusters": per.get("clusters"), "fragments": per.get("fragments"),
                   "front_end": steps[-1]["mode"]},
        "m2_setcoverfilter_wall_s": elapsed / K,
        "wall_s_per_step": {"clustering": per["cluster_s"], "filters (front end + MinHash NDF + scan + solve + strings out)": per["filter_s"],
                            "merge": per["merge_s"],
                            "clustering_stages": {k[len("cluster_"):]: per[k] for k in per if k.startswith("cluster_") and k != "cluster_s"}},
        "kernel_ms_per_step": {"k1_scan": per.get("scan_ms", 0.0), "k1_seed_verify": per.get("verify_ms", 0.0),
                               "k1_table_lookup": lookup_ms, "rows_build": per.get("rows_ms", 0.0),
                               "k2_greedy": per.get("greedy_ms", 0.0), "k2_greedy_rounds_only": per.get("rounds_ms", 0.0),
                               "k3_minhash_filter (two worker streams side by side: sums their stream times)": per.get("ndf_ms", 0.0)},
        "roofline": s5_roofline(dom, dms, dbytes, dl, args),
        "work_per_step": {"candidates": per.get("candidates"), "unique_candidates": per.get("unique_candidates"),
                          "table_matches": seeds, "hits": hits, "rows": rows, "picks": per.get("picks"),
                          "rounds": per.get("greedy_iters"), "probes": len(set(probes)),
                          "ndf_probes": N3, "ndf_tables": T3, "ndf_pairs_compared": C3, "ndf_kept": per.get("ndf_kept")},
        "probes_sha256": dg,
        "parity_vs_golden_digests": (None if gold is None else dg == gold["probes_sha256"]),
        "dataset_generation_s": gen_s,
        "step_seconds": [st["wall_s"] for st in steps],
        "device_memory": engine.pool_stats(),
    }
    if not args.no_cpu_baseline and args.scale == 1.0:
        base, gpu_s, same = cpu_baseline_s5()
        out["cpu_baseline"] = base
        out["parity_vs_oracle_on_cpu_sample"] = same
        out["gpu_ms_on_cpu_sample"] = gpu_s * 1e3
        out["speedup_vs_cpu_oracle_on_sample"] = base["seconds"] / gpu_s
    if not args.no_solver_check:
        # property check where no oracle digest exists: the two kernel families of the frontier solver
        # (set-parallel fused / row-parallel flat) must select the same probes
        cur_flat = per.get("flat_rows_streamed", 0) > 0
        checks = [] if not args.no_property_checks else None
        engine.collect_solution_checks(checks)
        try:
            p2, tm2 = one_step({"CATCHHIP_FLAT_MIN_ROWS": str(1 << 40) if cur_flat else "0"})
        finally:
            engine.collect_solution_checks(None)
        if checks is not None:
            # (the solutions replayed are those of this pass, the other solver family's; the digest below ties them to the timed ones)
            out["property_checks"] = summarize_checks(checks)
        out["solver_families_agree"] = probes_digest(p2) == dg
        out["solver_family_timed"] = "flat (row-parallel)" if cur_flat else "fused (set-parallel)"
        out["other_family_wall_s"] = tm2["wall_s"]
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="S4")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-m2", action="store_true", help="skip the M2 (cold, PCIe-inclusive) passes")
    ap.add_argument("--no-partial", action="store_true", help="skip the extra pass under -c 0.9")
    ap.add_argument("--m2-steps", type=int, default=3)
    ap.add_argument("--no-overlap-figure", action="store_true",
                    help="skip the extra passes with all groups on three streams (N = 1 only)")
    ap.add_argument("--groups-in-flight", type=int, default=4,
                    help="groups running at once per GPU, each on its own stream")
    ap.add_argument("--preflight", action="store_true",
                    help="one untimed-quality step on every rank: asserts the digests and prints per-rank "
                         "pack / scan / rows / solve / exchange times and the RCCL copy in use -- what to look "
                         "at first when a multi-GPU run misbehaves")
    ap.add_argument("--no-property-checks", action="store_true",
                    help="skip the extra step whose solutions are replayed by the independent check kernels")
    ap.add_argument("--no-also", action="store_true",
                    help="default S4 run: do not run configs[2] (S3) and configs[4] (S5) afterwards")
    ap.add_argument("--also-s5-scale", type=float, default=1.0)
    ap.add_argument("--no-solver-check", action="store_true",
                    help="S5: skip the extra pass through the other solver family")
    args = ap.parse_args()
    if args.preflight:
        args.steps, args.warmup = 1, 0
        args.no_cpu_baseline = args.no_m2 = args.no_partial = args.no_overlap_figure = True
    if args.workload in ("S5", "S5m"):
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("--workload S5 runs on one GPU (the clustered design is one process)")
        return bench_design_large(args)

    W = parallel.init_from_env()          # TCP process group (catch_amd.netstore) + RCCL communicator when WORLD_SIZE > 1
    rank, world, dist = W.rank, W.size, W.dist
    device = engine.default_context().device

    t_gen = time.perf_counter()
    groups = synthetic.dataset(args.workload, scale=args.scale)
    gen_s = time.perf_counter() - t_gen
    bases = [sum(len(s) for g in grp for s in g) for grp in groups]
    # groups above an even share of the work are sharded over all ranks by
    # universes (RCCL exchanges per solver round), the others go whole to ranks
    sharded_idx, plan = parallel.plan_with_sharding(
        bases, world, min_cost=int(os.environ.get("CATCHHIP_SHARD_MIN_BASES", "30000000")))
    mine = plan[rank]

    Stepper.ndf = args.workload == "S3"
    t_up0 = time.perf_counter()
    stepper = Stepper(device, groups, mine, args.groups_in_flight)
    sharded = [ShardedGroup(W, i, groups[i]) for i in sharded_idx]
    stepper.sync()
    upload_s = time.perf_counter() - t_up0
    units = sum(g.n_sets * g.G for g in stepper.resident) + sum(g.n_sets * g.G_local for g in sharded)
    n_cands = sum(g.n_sets for g in stepper.resident) + (sum(g.n_sets for g in sharded) if rank == 0 else 0)

    def run_step(stats=None, beside=True):
        out = {}
        for g in sharded:                 # all ranks together, in the same order
            out[g.index] = g.step(stats)
        out.update(stepper.step(stats, beside))
        return out

    def barrier():
        stepper.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        run_step()
    stats = []
    barrier()
    pool0 = engine.pool_stats()
    t0 = time.perf_counter()
    picks = None
    step_s = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        picks = run_step(stats)
        step_s.append(time.perf_counter() - ts)
    stepper.sync()
    elapsed = time.perf_counter() - t0
    pool1 = engine.pool_stats()
    my_elapsed = elapsed

    # one extra, untimed step with the E_dirty statistics on (SURVEY 8(d) K2's byte formula wants them), and one
    # whose solutions are replayed by the independent check kernels
    dirty_stats, prop = None, None
    alone_stats, alone_ms = None, None
    if world == 1 and not args.preflight and stepper.union_pool is not None:
        # the timed steps run two chains side by side (the large groups | the union instance): their per-kernel
        # event times include what the other stream's kernels took.  Two more untimed steps one chain after the
        # other give every kernel's time on an otherwise idle device -- reported beside the timed figures
        run_step(None, beside=False)
        alone_stats = []
        stepper.sync()
        ta = time.perf_counter()
        run_step(alone_stats, beside=False)
        stepper.sync()
        alone_ms = (time.perf_counter() - ta) * 1e3
    if world == 1 and not args.preflight:
        os.environ["CATCHHIP_FLAT_COUNT_DIRTY"] = "1"
        try:
            st_d = []
            run_step(st_d)
            dirty_stats = {"rows_recounted": float(sum(st.get("rows_recounted", 0) for st in st_d)),
                           "bitmap_words_read": float(sum(st.get("bitmap_words_read", 0) for st in st_d))}
        finally:
            os.environ.pop("CATCHHIP_FLAT_COUNT_DIRTY", None)
        if not args.no_property_checks:
            checks = []
            engine.collect_solution_checks(checks)
            try:
                run_step()
            finally:
                engine.collect_solution_checks(None)
            prop = summarize_checks(checks)
            if Stepper.ndf:
                import random
                from catch_amd.filter.near_duplicate_filter import NearDuplicateFilterWithHammingDistance
                g0 = stepper.resident[0]
                random.seed(7)
                tabs = NearDuplicateFilterWithHammingDistance(NdfGroup.HAMMING, PROBE_LEN)._draw_positions()
                flat = np.frombuffer("".join(s_ for g in groups[g0.index] for s_ in g).encode(), dtype=np.uint8)
                at = np.asarray(g0.cands.positions(), dtype=np.int64)
                kept = flat[at[:, None] + np.arange(PROBE_LEN)[None, :]]
                npairs, nbad = ndf_bucket_independence(kept, tabs, NdfGroup.HAMMING)
                prop = dict(prop or {}, ndf_kept=int(at.size), ndf_pairs_sharing_a_bucket=npairs, ndf_pairs_too_close=nbad)
                prop["ok"] = bool(prop.get("ok", True) and nbad == 0)
    # identical picks whatever N: every group against the committed digests
    gold = golden_digests(args.workload, args.scale)
    gold_ok, gold_n = True, 0
    if gold is not None:
        for gi, ids in picks.items():
            if gi in sharded_idx and rank != 0:
                continue                  # identical on every rank: counted once
            if gi in gold:
                gold_n += 1
                gold_ok = gold_ok and (len(ids) == gold[gi]["n_picks"] and
                                       digest(ids) == gold[gi]["picks_sha256"] and
                                       ("picks_in_order_sha256" not in gold[gi] or
                                        digest_in_order(ids) == gold[gi]["picks_in_order_sha256"]))
    total_units, all_elapsed = float(units), [elapsed]
    preflight = None
    if args.preflight:
        mine_tot = {}
        for st in stats:
            for k, v in st.items():
                if isinstance(v, (int, float)):
                    mine_tot[k] = mine_tot.get(k, 0) + v
        me = dict(rank=rank, device=device, groups=[g.index for g in stepper.resident],
                  sharded_groups=list(sharded_idx), pack_h2d_s=upload_s, step_s=elapsed,
                  scan_ms=mine_tot.get("scan_ms", 0.0), rows_ms=mine_tot.get("rows_ms", 0.0),
                  greedy_ms=mine_tot.get("greedy_ms", 0.0),
                  sharded_scan_wall_ms=mine_tot.get("sharded_scan_wall_ms", 0.0),
                  sharded_solve_wall_ms=mine_tot.get("sharded_solve_wall_ms", 0.0),
                  exchange="rccl" if W.rccl else ("gloo through host memory" if world > 1 else "none"),
                  rccl=engine.Context.comm_info(), digests_ok=bool(gold_ok), groups_checked=gold_n)
        preflight = W.allgather(me)
    if dist is not None:
        # (host objects over the process group: catch_amd.netstore -- no torch)
        parts = W.allgather((my_elapsed, float(units), float(gold_n), 0.0 if gold_ok else 1.0, float(n_cands)))
        elapsed = max(p_[0] for p_ in parts)
        total_units = sum(p_[1] for p_ in parts)
        gold_n = int(sum(p_[2] for p_ in parts))
        gold_ok = all(p_[3] == 0.0 for p_ in parts)
        n_cands = int(sum(p_[4] for p_ in parts))
        all_elapsed = [p_[0] for p_ in parts]
        W.barrier()

    if rank == 0:
        K = args.steps
        P_all = n_cands
        P = sum(g.n_sets for g in stepper.resident)
        G = sum(g.G for g in stepper.resident)
        tot = {}
        for st in stats:
            for k, v in st.items():
                tot[k] = tot.get(k, 0) + v
        per = {k: v / K for k, v in tot.items()}        # per step, this rank
        seeds, hits, rows = per.get("seed_hits", 0), per.get("raw_hits", 0), per.get("rows", 0)
        # ---- bytes per step ("Roofline accounting") ----
        # K1 since round 4 = the key-grouped join (csrc/scan_join.inc).  H = target positions whose k-mer is in the
        # anchor table, pairs = (position, entry) pairs verified, E = anchor entries in the table (m + 1 per candidate).
        H, pairs = per.get("join_hit_positions", 0), per.get("join_pairs", 0)
        E = float(MISMATCHES + 1) * P
        dropped = per.get("seeds_dropped", 0)
        live = seeds - dropped
        # one pass of kj_verify: per hit position a 12-B record + two sequence bounds + (m + 1) windows of
        # (NW + 1) 16-byte words; per table entry its id, 64-byte probe image, bucket and offset; the counting pass
        # writes 4 B per entry, the writing pass a 16-B record per hit
        nw = -(-PROBE_LEN // 32)
        pass_bytes = H * (12 + 8 + (MISMATCHES + 1) * 16.0 * (nw + 1)) + E * (4 + 16.0 * nw + 8)
        verify_bytes = 2 * pass_bytes + 4.0 * E + 16.0 * hits
        # SURVEY 8(d) K1 (brute-force tiles, T_p = 1024) for the same groups
        survey_k1 = sum(0.375 * g.G * -(-g.n_sets // 1024) + 0.375 * PROBE_LEN * g.n_sets
                        for g in stepper.resident) + 16.0 * rows
        # whole scan phase as implemented: table build (per entry: 64-B image read, 16-B slot, 4-B list entry, twice
        # its slot number), hit positions (planes 0/1 = 0.25 B per base + a presence bit; 12 B staged, read and
        # written again per hit position), 4 radix passes of 12 B in and out, the two verify passes
        k1_impl = 100.0 * E + 0.25 * G + 36.0 * H + 4 * 24.0 * H + verify_bytes
        rows_bytes = 32.0 * hits + 32.0 * rows      # merge: records in, merged rows out; emit: 16 B in, 16 B out
        # 12 B per (set, universe, interval) row of E_dirty + 8 B per bitmap word read for the
        # popcounts.  E_dirty (rows that hold a word the previous picks changed) is counted in one extra, untimed
        # step (CATCHHIP_FLAT_COUNT_DIRTY); the rows the launches actually count again are all alive ones.
        edirty = dirty_stats.get("rows_recounted") if dirty_stats else None
        ewords = dirty_stats.get("bitmap_words_read") if dirty_stats else None
        k2_bytes = (12.0 * (edirty if edirty is not None else per.get("rows_recounted", 0))
                    + 8.0 * (ewords if ewords is not None else per.get("bitmap_words_read", 0)))
        k2_full = 12.0 * per.get("rows_recounted", 0) + 8.0 * per.get("bitmap_words_read", 0)
        # what the rounds move as implemented: every alive record read by the count launch (8 B), written for every
        # survivor (8 B), read again by the claim launch (8 B), one bitmap word per flagged word and one owner word
        streamed = per.get("flat_rows_streamed", 0)
        k2_impl = 24.0 * streamed + 8.0 * per.get("bitmap_words_read", 0) + 8.0 * per.get("flat_owner_words", 0)
        ms = {k: per.get(k, 0.0) for k in ("scan_ms", "verify_ms", "vcount_ms", "rows_ms", "greedy_ms", "rounds_ms", "claim_ms")}
        nlaunch = {k: tot.get(k, 0) for k in ("verify_launches", "vcount_launches", "rounds_launches", "scan_launches", "claim_launches")}

        def gbs(b, t_ms):
            return b / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
        ninst = float(len(stepper.big_alone) + 1 if stepper.union is not None else len(stepper.resident))
        units_roof = {
            "join_verify": dict(kernel="kj_verify_kernel<%d, true/false> + kj_giant_kernel (key-grouped join: counting + writing pass)" % nw,
                                ms=ms["verify_ms"] + ms["vcount_ms"], bytes=verify_bytes,
                                launches=(nlaunch["verify_launches"] + nlaunch["vcount_launches"]) / K, pmc="join_verify"),
            "solver_claim": dict(kernel="gr_claim_kernel (row-parallel solver, instances of >= 262,144 rows)",
                                 ms=ms["claim_ms"], bytes=per.get("claim_bytes", 0.0),
                                 launches=max(nlaunch["claim_launches"], 1) / K, pmc="gr_claim"),
            "rows_build": dict(kernel="bucketed row build (per group: merge, scans, emit)",
                               ms=ms["rows_ms"], bytes=rows_bytes, launches=ninst, pmc="rows_build"),
            # the K2 UNIT: a frontier round = gr_count + gr_claim + gr_apply + gr_cover, priced with
            # SURVEY 8(d)'s formula (12 B per row of E_dirty + 8 B per bitmap word read); a "launch" is one round
            "k2_rounds": dict(kernel="frontier solver round (gr_count + gr_claim + gr_apply + gr_cover; SURVEY 8(d) K2 bytes)",
                              ms=ms["rounds_ms"], bytes=k2_bytes, launches=max(per.get("greedy_iters", 0), 1), pmc="solver_round"),
        }
        ta_ = {}
        for st in (alone_stats or []):
            for k, v in st.items():
                ta_[k] = ta_.get(k, 0) + v
        alone_ms_of = {"join_verify": ta_.get("verify_ms", 0.0) + ta_.get("vcount_ms", 0.0), "solver_claim": ta_.get("claim_ms", 0.0),
                       "rows_build": ta_.get("rows_ms", 0.0), "k2_rounds": ta_.get("rounds_ms", 0.0)}

        def roof_of(name):
            d = units_roof[name]
            avg_ms = d["ms"] / max(d["launches"], 1)
            traffic = pmc_traffic(d["pmc"], args.workload, args.scale)
            alg_pl = d["bytes"] / max(d["launches"], 1)
            # `achieved` never prices more bytes than were counted on the memory side: min(model, PMC traffic)
            priced = alg_pl if traffic is None else min(alg_pl, traffic)
            roof = dict(bound="hbm", kernel=d["kernel"], achieved=gbs(priced, avg_ms),
                        peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs(priced, avg_ms) / HBM_PEAK_GBS,
                        traffic=traffic,
                        algorithmic_bytes_per_launch=alg_pl,
                        priced_bytes_per_launch=priced,
                        avg_launch_ms=avg_ms, launches_per_step=d["launches"],
                        device_ms_per_step=d["ms"],
                        note="frac = min(algorithmic bytes, PMC traffic) per launch / HIP-event time per launch / peak")
            rec = _pmc_record(args.workload, args.scale)
            if rec is not None:
                roof["traffic_source"] = rec["_file"]
                roof["traffic_source_round"] = rec["_round"]
            if alone_stats:
                a_ms = alone_ms_of[name]
                a_avg = a_ms / max(d["launches"], 1)
                # The line's roofline is priced with the unit's launches when they have the device to themselves (VERDICT
                # round 4): the timed steps overlap two chains, so a launch's event time there includes the other stream's
                # kernels.  The overlapped figures stay beside it.
                roof["overlapped_in_timed_steps"] = dict(avg_launch_ms=roof["avg_launch_ms"], device_ms_per_step=roof["device_ms_per_step"],
                                                         achieved=roof["achieved"], frac=roof["frac"])
                roof.update(avg_launch_ms=a_avg, device_ms_per_step=a_ms, achieved=gbs(priced, a_avg), frac=gbs(priced, a_avg) / HBM_PEAK_GBS)
                roof["note"] += ("; time = the unit's launches in an untimed step that runs the union instance AFTER the large "
                                 "groups instead of beside them (one chain at a time: what `rocprofv3 --kernel-trace --stats` of "
                                 "CATCHHIP_BENCH_UNION_BESIDE=0 sums, profiles/r<NN>_bench_kernel_stats_one_chain.csv); "
                                 "overlapped_in_timed_steps: the same from the timed steps' event times")
            return roof
        # The dominant UNIT by its own HIP-event time: the solver's rounds (K2), the join's verify launches, or the row
        # build -- the K2 unit since round 2.  The claim kernel alone (rounds 2-5's `roofline`) stays beside it.
        dom = max((k for k in units_roof if k != "solver_claim"), key=lambda k: units_roof[k]["ms"])
        roof = roof_of(dom)
        roof_claim = roof_of("solver_claim")
        d = units_roof[dom]
        alone = None
        if alone_stats:
            alone = {"ms_per_step": alone_ms,
                     "kernel_ms_per_step": {"k1_scan": ta_.get("scan_ms", 0.0), "k1_join_verify_count": ta_.get("vcount_ms", 0.0),
                                            "k1_join_verify_write": ta_.get("verify_ms", 0.0), "rows_build": ta_.get("rows_ms", 0.0),
                                            "k2_greedy": ta_.get("greedy_ms", 0.0), "k2_greedy_rounds_only": ta_.get("rounds_ms", 0.0),
                                            "k2_claim_launches": ta_.get("claim_ms", 0.0)}}
        out = {
            "metric": "candidate-probe x target-bp / s through SetCoverFilter "
                      "(K1 scan + K2 greedy)",
            "value": total_units * K / elapsed,
            "unit": "probe*bp/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "u32 bit-planes / u64 bitmap (integer)",
            "data": "synthetic",
            "config": {"workload": "%s (BASELINE configs[%d]%s): %d genomes in "
                                   "%d groups, G=%d bp, P=%d candidates, "
                                   "-pl 100 -ps 50 -m 2 -e 50 -c 1.0"
                                   % (args.workload, CONFIG_OF.get(args.workload, 1),
                                      "" if args.scale == 1.0 else " scaled x%g" % args.scale,
                                      sum(len(g) for g in groups), len(groups),
                                      sum(bases), P_all),
                       "shard": "groups above an even share of the bases are sharded over all ranks by "
                                "universes (2 RCCL all-reduces per solver round), the others go whole to "
                                "ranks, longest first (no collective)",
                       "sharded_groups": list(sharded_idx),
                       "groups_on_rank0": [g.index for g in stepper.resident],
                       "groups_in_flight": stepper.width, "scale": args.scale,
                       "one_instance": (stepper.union.indices if stepper.union is not None else []),
                       "one_instance_note": "the groups below %g Mbases are scanned and solved as ONE instance (targets / "
                                            "candidates / probes that carry group numbers); every group's picks and "
                                            "their order are its own and are checked against its own digests"
                                            % (Stepper.UNION_BELOW / 1e6)},
            "setcoverfilter_ms": elapsed / K * 1e3,
            "picks": per.get("picks", 0), "rows": rows,
            "kernel_ms_per_step": {"k1_scan": ms["scan_ms"], "k1_join_verify_count": ms["vcount_ms"],
                                   "k1_join_verify_write": ms["verify_ms"],
                                   "rows_build": ms["rows_ms"], "k2_greedy": ms["greedy_ms"],
                                   "k2_greedy_rounds_only": ms["rounds_ms"], "k2_claim_launches": ms["claim_ms"],
                                   "note": "HIP-event device time summed over the groups of rank 0; "
                                           "groups in flight overlap, so the sum can exceed ms_per_step"},
            "one_chain_at_a_time": alone,
            "roofline": roof,
            "roofline_claim_kernel": roof_claim,
            "roofline_k1_verify": dict(bound="hbm", achieved=gbs(verify_bytes, ms["verify_ms"] + ms["vcount_ms"]),
                                       peak=HBM_PEAK_GBS, unit="GB/s",
                                       frac=gbs(verify_bytes, ms["verify_ms"] + ms["vcount_ms"]) / HBM_PEAK_GBS,
                                       implementation_bytes=verify_bytes,
                                       traffic=pmc_traffic("join_verify", args.workload, args.scale),
                                       valu_note="the verify passes are integer-issue work, not bandwidth: %.3g pairs x ~45 "
                                                 "lane-ops per pass" % pairs,
                                       lane_utilisation=(pairs / per["join_lane_slots"]) if per.get("join_lane_slots") else None),
            "roofline_k1_scan": dict(bound="hbm", implementation_bytes=k1_impl,
                                     achieved_implementation=gbs(k1_impl, ms["scan_ms"]),
                                     frac_implementation=gbs(k1_impl, ms["scan_ms"]) / HBM_PEAK_GBS,
                                     survey_8d_formula_bytes=survey_k1,
                                     achieved_survey_formula=gbs(survey_k1, ms["scan_ms"]),
                                     peak=HBM_PEAK_GBS, unit="GB/s",
                                     note="SURVEY 8(d)'s K1 bytes price a brute-force tiled scan "
                                          "(every probe tile re-reads the targets); the join "
                                          "does not do that work, so that figure is an upper "
                                          "courtesy, not an achievement"),
            "roofline_rows": dict(bound="hbm", achieved=gbs(rows_bytes, ms["rows_ms"]),
                                  peak=HBM_PEAK_GBS, unit="GB/s",
                                  frac=gbs(rows_bytes, ms["rows_ms"]) / HBM_PEAK_GBS,
                                  implementation_bytes=rows_bytes,
                                  traffic=pmc_traffic("rows_build", args.workload, args.scale)),
            "roofline_k2": dict(bound="hbm", achieved=gbs(k2_bytes, ms["rounds_ms"]),
                                peak=HBM_PEAK_GBS, unit="GB/s",
                                frac=gbs(k2_bytes, ms["rounds_ms"]) / HBM_PEAK_GBS,
                                algorithmic_bytes=k2_bytes, e_dirty_rows=edirty, e_dirty_words=ewords,
                                full_recompute_bytes=k2_full,
                                traffic=pmc_traffic("solver_round", args.workload, args.scale),
                                implementation_bytes=k2_impl,
                                achieved_implementation=gbs(k2_impl, ms["rounds_ms"]),
                                note="algorithmic_bytes = SURVEY 8(d) K2: 12 B per row of E_dirty (a word of it changed since "
                                     "the last round; counted in one extra untimed step) + 8 B per bitmap word read; "
                                     "full_recompute_bytes prices every alive row (what the count launches do since round 4); "
                                     "implementation_bytes adds the record streams (count in and out, claim in) and the owner words",
                                us_per_pick=ms["rounds_ms"] * 1e3 / max(per.get("picks", 0), 1),
                                rounds_per_step=per.get("greedy_iters", 0)),
            "roofline_k3": None,
            "traffic_per_step": traffic_per_step(args.workload, args.scale),
            "work_per_step": {"hit_positions": H, "table_matches": pairs, "pairs_verified": pairs, "hits": hits, "rows": rows,
                              "rows_recounted": per.get("rows_recounted", 0), "e_dirty_rows": edirty,
                              "bitmap_words_read": per.get("bitmap_words_read", 0),
                              "flat_rows_streamed": streamed,
                              "flat_owner_words": per.get("flat_owner_words", 0)},
            "dataset_generation_s": gen_s,
            "first_upload_s": upload_s,
            "rank_seconds": all_elapsed,
            "step_seconds_rank0": step_s,
            "device_memory": dict(pool1, hipmalloc_calls_in_timed_region=(
                pool1["hipmalloc_calls"] - pool0["hipmalloc_calls"])),
            "parity_vs_golden_digests": (gold_ok if gold_n else None),
            "groups_checked_against_digests": gold_n,
            "property_checks": prop,
            "rccl": engine.Context.comm_info() if world > 1 else None,
        }

        if alone is not None:
            akm = alone["kernel_ms_per_step"]
            out["roofline_k2"]["frac_one_chain_at_a_time"] = gbs(k2_bytes, akm["k2_greedy_rounds_only"]) / HBM_PEAK_GBS
            out["roofline_rows"]["frac_one_chain_at_a_time"] = gbs(rows_bytes, akm["rows_build"]) / HBM_PEAK_GBS
            out["roofline_k1_verify"]["frac_one_chain_at_a_time"] = gbs(
                verify_bytes, akm["k1_join_verify_count"] + akm["k1_join_verify_write"]) / HBM_PEAK_GBS
        if preflight is not None:
            out["preflight"] = preflight
        if Stepper.ndf:
            # N (ceil(0.375 L) + T (8 + 4) passes) + 2 ceil(0.375 L) C -- N probes, T tables,
            # 8 radix passes over 64-bit keys + 32-bit indices, C pairs sharing a bucket that were compared
            N, T, C = per.get("ndf_probes", 0), per.get("ndf_tables", 0) / max(len(stepper.resident), 1), per.get("ndf_pairs", 0)
            lb = -(-3 * PROBE_LEN // 8)
            k3_bytes = N * (lb + T * 12 * 8) + 2.0 * lb * C
            out["roofline_k3"] = dict(bound="hbm", kernel="ndf_key + radix sort + ndf_edge + resolution rounds (Hamming LSH filter)",
                                      achieved=gbs(k3_bytes, per.get("ndf_ms", 0.0)), peak=HBM_PEAK_GBS, unit="GB/s",
                                      frac=gbs(k3_bytes, per.get("ndf_ms", 0.0)) / HBM_PEAK_GBS,
                                      traffic=pmc_traffic("ndf", args.workload, args.scale),
                                      algorithmic_bytes=k3_bytes, device_ms_per_step=per.get("ndf_ms", 0.0),
                                      probes=N, tables=T, pairs_compared=C, edges=per.get("ndf_edges", 0),
                                      kept=per.get("ndf_kept", 0), windows=per.get("windows", 0),
                                      front_end_ms=per.get("front_end_ms", 0.0))
            urec = units_record(args.workload, args.scale)
            if urec is not None:
                # the filter is not HBM-bound (0.03 of the HBM peak says nothing): its VALU side from the counters
                vf = valu_figures(urec["units"].get("ndf"), urec.get("steps_in_run", 1), per.get("ndf_ms", 0.0))
                if vf is not None:
                    out["roofline_k3"] = closer_roof(out["roofline_k3"], vf)
            out["config"]["workload"] += "; step = device front end + --filter-with-lsh-hamming 2 (K3) + scan + solve, targets resident"
            if per.get("ndf_ms", 0.0) > d["ms"]:
                # configs[2]: the filter is the dominant unit of the step (one filter call per group = one "launch")
                nf = float(len(stepper.resident))
                k3 = out["roofline_k3"]
                out["roofline"] = dict(bound="hbm", kernel=k3["kernel"], achieved=k3["achieved"], peak=HBM_PEAK_GBS,
                                       unit="GB/s", frac=k3["frac"], traffic=k3["traffic"],
                                       algorithmic_bytes_per_launch=k3_bytes / nf,
                                       avg_launch_ms=per.get("ndf_ms", 0.0) / nf, launches_per_step=nf,
                                       device_ms_per_step=per.get("ndf_ms", 0.0))
        if world == 1 and not args.no_cpu_baseline and Stepper.ndf:
            # configs[2]: the oracle's chain (Hamming filter, then the set cover) on a down-scaled S3 of the
            # same generator, and the GPU on that same input
            base, gpu_s, same = cpu_baseline_s3(stepper.ctxs[0])
            out["cpu_baseline"] = base
            out["parity_vs_oracle"] = same
            out["gpu_ms_on_cpu_sample"] = gpu_s * 1e3
            out["speedup_vs_cpu_oracle"] = base["seconds"] / gpu_s
        elif world == 1 and not args.no_cpu_baseline:
            order = sorted(range(len(groups)), key=lambda i: bases[i])
            by_idx_ok = set(g.index for g in stepper.resident)
            sample, acc = [], 0
            for i in order:
                if sample and acc + bases[i] > 13_000_000:
                    break
                sample.append(i)
                acc += bases[i]
            mids = [i for i in range(len(groups)) if 20_000_000 <= bases[i] <= 40_000_000 and i in by_idx_ok]
            mid = min(mids, key=lambda i: abs(bases[i] - 30_000_000)) if mids else None
            base, sel = cpu_baseline(groups, sample, mid=mid)
            out["cpu_baseline"] = base
            checked = sample + ([mid] if mid is not None else [])
            # the timed GPU result must equal the oracle's (parity guard)
            out["parity_vs_oracle"] = all(sorted(int(x) for x in picks[gi]) == sorted(sel[gi])
                                          for gi in checked)
            # like for like: the GPU on exactly the CPU sample's groups (resident inputs, one
            # after the other), against the CPU's time per pass over the same groups
            by_index = {g.index: g for g in stepper.resident}

            def gpu_seconds(gis, reps=5):
                for _ in range(2):
                    for gi in gis:
                        g = by_index[gi]
                        engine.setcover_filter(g.ctx, g.probes, g.targets, MISMATCHES, PROBE_LEN, 0, EXT,
                                               g.n_sets, mode=SCAN_MODE)
                stepper.sync()
                tg = time.perf_counter()
                for _ in range(reps):
                    for gi in gis:
                        g = by_index[gi]
                        engine.setcover_filter(g.ctx, g.probes, g.targets, MISMATCHES, PROBE_LEN, 0, EXT,
                                               g.n_sets, mode=SCAN_MODE)
                stepper.sync()
                return (time.perf_counter() - tg) / reps
            gpu_sample_s = gpu_seconds(sample)
            out["gpu_ms_on_cpu_sample"] = gpu_sample_s * 1e3
            out["speedup_vs_cpu_oracle"] = (base["seconds"] / base["passes"]) / gpu_sample_s
            if mid is not None:
                gm = gpu_seconds([mid], reps=3)
                base["mid_group"]["gpu_ms"] = gm * 1e3
                base["mid_group"]["speedup_vs_cpu_oracle"] = base["mid_group"]["seconds"] / gm
            out["speedup_note"] = ("CPU seconds per pass over the sample's groups / GPU seconds for the same "
                                   "groups (resident inputs); `value` / cpu_baseline.value is NOT comparable: "
                                   "probe*bp grows quadratically with the group size")
        if world == 1 and not args.no_partial and not Stepper.ndf and args.workload in ("S4", "S4i"):
            # the same resident groups under -c 0.9 (partial coverage): frontier rounds with the universe
            # test; digests of the picks in order against the committed ones
            def c09_pass(collect):
                # (the same instances as the timed step: the large groups one by one, the small ones as their union)
                out09, st09 = {}, dict(greedy_ms=0.0, rounds=0, picks=0)
                u = stepper.union
                for g in (stepper.big_alone + [u] if u is not None else stepper.resident):
                    ngen = g.n_genomes if g is not u else int(g.targets.ngenomes)
                    ids, _ = engine.setcover_filter(g.ctx, g.probes, g.targets, MISMATCHES, PROBE_LEN, 0, EXT, g.n_sets,
                                                    universe_p=[0.9] * ngen, mode=SCAN_MODE)
                    if g is u:
                        out09.update(u.split(ids))
                    else:
                        out09[g.index] = ids
                    if collect:
                        st09["greedy_ms"] += g.ctx.kernel_ms(engine.PHASE_GREEDY)[0]
                        st09["rounds"] += g.ctx.counters()["greedy_iters"]
                        st09["picks"] += len(ids)
                return out09, st09
            c09_pass(False)
            stepper.sync()
            t9 = time.perf_counter()
            p09, st09 = c09_pass(True)
            stepper.sync()
            el9 = time.perf_counter() - t9
            ok9 = None
            if gold is not None and all("picks_c09_sha256" in gold[gi] for gi in p09 if gi in gold):
                ok9 = all(len(ids) == gold[gi]["n_picks_c09"] and digest(ids) == gold[gi]["picks_c09_sha256"]
                          and digest_in_order(ids) == gold[gi]["picks_c09_in_order_sha256"]
                          for gi, ids in p09.items() if gi in gold)
            out["partial_coverage"] = {"coverage": 0.9, "ms_per_step": el9 * 1e3, "k2_greedy_ms": st09["greedy_ms"],
                                       "k2_greedy_ms_full_coverage": (alone["kernel_ms_per_step"]["k2_greedy"] if alone is not None else ms["greedy_ms"]),
                                       "rounds": st09["rounds"],
                                       "picks": st09["picks"], "parity_vs_golden_digests": ok9,
                                       "note": "one pass over the resident instances of the timed step (large groups one after "
                                               "the other, the small ones as one instance); digests are of every group's "
                                               "picks in its own pick order"}
        if world == 1 and not args.no_m2 and not Stepper.ndf:
            engine.pool_trim()        # (the resident passes' cached blocks: the passes below allocate on other contexts)
            # M2: from host strings to ids on the host, nothing resident, through the
            # plugin's pipelined path; then one pass without the overlap for comparison
            tm2, picks_m2 = m2_passes(groups, args.m2_steps, 1, int(os.environ.get("CATCHHIP_PREFETCH_DEPTH", "2")))
            m2 = sum(tm2) / len(tm2)
            ts2, picks_s2 = m2_passes(groups, 1, 1, 0)     # (one warm pass: its contexts' blocks were returned by the trim above)
            out["m2_setcoverfilter_wall_s"] = m2
            out["m2_steps_s"] = tm2
            out["m2_serial_wall_s"] = ts2[0]
            out["value_incl_h2d"] = total_units / m2
            out["m2_parity_vs_golden_digests"] = (None if gold is None else
                                                  bool(digests_ok(gold, picks_m2) and digests_ok(gold, picks_s2)))
            out["m2_note"] = ("pack + H2D + device front end + scan + solve + ids out per pass, host strings in, "
                              "via SetCoverFilter._filter_genomes_device; the upload context packs group i+1 "
                              "while group i is scanned and solved (m2_serial_wall_s: the same without overlap)")
        if world == 1 and not args.no_overlap_figure and not args.no_cpu_baseline:   # (both extras off under the profiler)
            # The same work with every group on one of three streams (what the plugin's
            # CATCHHIP_GROUPS_IN_FLIGHT does): faster, but kernels of different groups then
            # share the CUs and their HIP-event times stop meaning anything -- which is why
            # the line above keeps the large groups one after the other.
            stepper.close()
            stepper = None
            engine.pool_trim()
            Stepper.BIG_BASES = 1 << 62
            Stepper.UNION_BELOW = 0               # (every group on its own, three at a time)
            st2 = Stepper(device, groups, mine, 3)
            st2.sync()
            for _ in range(2):
                st2.step()
            st2.sync()
            t2 = time.perf_counter()
            picks2 = None
            for _ in range(3):
                picks2 = st2.step()
            st2.sync()
            el2 = (time.perf_counter() - t2) / 3
            ok2 = None
            if gold is not None:
                ok2 = bool(digests_ok(gold, picks2))
            out["groups_overlapped"] = {"groups_in_flight": st2.width, "steps": 3, "ms_per_step": el2 * 1e3,
                                        "value": total_units / el2, "parity_vs_golden_digests": ok2,
                                        "note": "not the headline: per-kernel times (roofline) are only "
                                                "meaningful when kernels run alone"}
            st2.close()
        if (world == 1 and args.workload == "S4" and args.scale == 1.0 and not args.no_also and not args.no_cpu_baseline
                and not args.preflight):
            # BASELINE configs[2] (S3: --filter-with-lsh-hamming in the step) and configs[4] (S5: the design_large chain)
            # in the same driver-run line: each in a process of its own, after this one has given its memory back
            import subprocess
            if stepper is not None:
                stepper.close()
                stepper = None
            engine.pool_trim()
            keep = ("ms_per_step", "value", "unit", "steps", "warmup", "kernel_ms_per_step", "roofline", "roofline_k3",
                    "wall_s_per_step", "parity_vs_golden_digests", "property_checks", "solver_families_agree",
                    "work_per_step", "dataset_generation_s", "probes_sha256", "m2_setcoverfilter_wall_s", "cpu_baseline",
                    "parity_vs_oracle_on_cpu_sample", "gpu_ms_on_cpu_sample", "speedup_vs_cpu_oracle_on_sample")
            also = {}
            for name, extra in (("S3", ["--workload", "S3", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]),
                                # (a cold first step is 3-4 s longer: first-use allocations of 100+ GB; the S5 leg times its
                                # own CPU sample -- the oracle's MinHash filter + set cover on 40 genomes, ~20 s)
                                ("S5", ["--workload", "S5", "--scale", "%g" % args.also_s5_scale, "--steps", "2", "--warmup", "1"])):
                ta = time.perf_counter()
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra +
                                       ["--no-m2", "--no-partial", "--no-overlap-figure", "--no-also"],
                                       capture_output=True, text=True, timeout=1200)
                    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
                    sub = json.loads(line)
                    rec = {k: sub[k] for k in keep if k in sub}
                    rec["workload"] = sub["config"]["workload"]
                    rec["bytes_held"] = sub.get("device_memory", {}).get("bytes_held")
                except Exception as e:      # noqa: BLE001 -- the S4 line must not be lost to a failure here
                    rec = {"error": "%s: %s" % (type(e).__name__, e)}
                rec["wall_s"] = time.perf_counter() - ta
                also[name] = rec
            out["also"] = also
        print(json.dumps(out))
    for g in sharded:
        g.close()
    if stepper is not None:
        stepper.close()
    W.close()


if __name__ == "__main__":
    main()
