"""Host-side partitioning for the multi-GPU paths (pure Python: no device, no
torch -- so the world-size-2 gloo tests drive exactly this code).

Level 1, whole groups: the groups of a filter call are independent set-cover
instances (catch/filter/set_cover_filter.py:816-846); the reference hands
them to its pool largest first (:880-887).  `shard_plan` does the same over
ranks: longest-processing-time-first onto the least loaded rank.

Level 2, inside one group (catch_amd/csrc/setcover_sharded.inc): the group's
genomes (universes) are cut into contiguous ranges of roughly equal bases,
one per rank -- `split_universes`; every rank scans all candidates against
its genomes and the frontier solver exchanges one SUM all-reduce of the
per-set gains and one MAX all-reduce of the per-set "lost a word" flags per
round.  `merge_picks` restores the sequential pick order from the accepted
(set, key) pairs, which are identical on every rank.
"""
import heapq


def lpt_assign(costs, nbins):
    """Longest-processing-time-first: items (by index) in descending cost go to
    the currently least loaded bin.  Returns nbins lists of indices, each in
    the order its items should run (largest first).  Deterministic: ties by
    index."""
    nbins = max(1, int(nbins))
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    heap = [(0, b) for b in range(nbins)]
    bins = [[] for _ in range(nbins)]
    for i in order:
        load, b = heapq.heappop(heap)
        bins[b].append(i)
        heapq.heappush(heap, (load + costs[i], b))
    return bins


def shard_plan(group_costs, world):
    """Whole groups onto `world` ranks, largest first (level 1)."""
    return lpt_assign(list(group_costs), world)


def split_universes(genome_lengths, world):
    """Contiguous genome ranges [g0, g1) per rank with about equal bases
    (level 2).  Returns world+1 boundaries; a rank may get an empty range."""
    total = sum(genome_lengths)
    bounds, acc, g = [0], 0, 0
    for r in range(1, world):
        want = total * r / world
        while g < len(genome_lengths) and acc + genome_lengths[g] / 2.0 <= want:
            acc += genome_lengths[g]
            g += 1
        bounds.append(g)
    bounds.append(len(genome_lengths))
    return bounds


def merge_picks(picks, keys, ranks=None):
    """Sequential pick order of the frontier solver's accepted sets: by rank,
    then by descending accept-time key (setcover_batched.inc)."""
    idx = sorted(range(len(picks)),
                 key=lambda i: ((ranks[picks[i]] if ranks is not None else 0),
                                -keys[i]))
    return [picks[i] for i in idx]


# ---------------------------------------------------------------------------
# the round loop of a universe-sharded solve (include/catchhip.h,
# catchhip_shard_*).  `shard` is anything with count() / claim_check() /
# apply() / picks(); `exchange(which)` performs the all-reduce of the gain
# (which = 0, SUM) or lost (which = 1, MAX) buffers of all shards of the
# instance.  The same loop serves one process per GPU (RCCL), several shards in
# one process (tests on one GPU) and the CPU stand-ins of the gloo tests.
# ---------------------------------------------------------------------------
def sharded_solve(shards, exchange, native=None):
    """Runs the rounds of one instance over the shards THIS process holds
    (normally one) and returns the picks in the sequential pick order.
    Raises IndexError-like CatchHipError from picks() when the rank list is
    exhausted, as the unsharded solver does.
    native: "rccl" (one shard per process, a communicator on its context) or
    "local" (the shards of this process, one context): the loop runs under the
    C ABI (catchhip_shard_solve, round 6: several rounds per host read-back)
    and `exchange` is not used; None: the loop below, one library call per
    step and whatever transport `exchange` is (the host / TCP fallback, the
    CPU stand-ins of the tests)."""
    if native is not None:
        from catch_amd import engine
        return engine.shards_solve(shards, native)
    while True:
        for sh in shards:
            sh.count()
        exchange(0)
        for sh in shards:
            sh.claim_check()
        exchange(1)
        # partial coverage (every rank alike: some universe of the INSTANCE may stay partly uncovered): the
        # candidates that lost nowhere take the universe test on every rank, failures travel as lost marks
        if any(getattr(sh, "partial_instance", getattr(sh, "partial", False)) for sh in shards):
            for sh in shards:
                sh.verdict()
            exchange(1)
        done = [sh.apply() for sh in shards]
        if any(d != done[0] for d in done):
            raise RuntimeError("sharded solve: shards disagree on termination")
        if done[0]:
            break
    out = [sh.picks() for sh in shards]
    if any(o != out[0] for o in out):
        raise RuntimeError("sharded solve: shards returned different picks")
    return out[0]


def plan_with_sharding(group_costs, world, min_cost=0, slack=1.08, eligible=None):
    """Two-level plan.  Groups are SHARDED over all ranks (every rank takes
    1/world of each) until the others -- whole groups, longest first onto the
    least loaded rank -- leave the busiest rank within `slack` of an even share:
    the largest whole group of at least min_cost is sharded next as long as
    they do not.  (A group above an even share always ends up sharded; so does
    e.g. S4's 265-Mbase group on 2 ranks, 45 % of the work, which whole would
    leave the ranks at 265 : 327.)
    eligible: per group, whether it may be sharded at all (e.g. it has at least
    as many genomes as there are ranks: a rank without a genome would hold an
    empty shard); None = every group.
    Returns (sharded: list of group indices, whole: world lists of indices)."""
    costs = list(group_costs)
    if world <= 1:
        return [], [sorted(range(len(costs)), key=lambda i: (-costs[i], i))]
    even = sum(costs) / float(world)
    sharded = set()
    while True:
        rest = [i for i in range(len(costs)) if i not in sharded]
        order = sorted(rest, key=lambda i: (-costs[i], i))
        base = sum(costs[i] for i in sharded) / float(world)
        heap = [(base, b) for b in range(world)]
        bins = [[] for _ in range(world)]
        for i in order:
            load, b = heapq.heappop(heap)
            bins[b].append(i)
            heapq.heappush(heap, (load + costs[i], b))
        busiest = max(load for load, _ in heap)
        cand = [i for i in order if costs[i] >= min_cost and costs[i] > 0
                and (eligible is None or eligible[i])]
        if not cand or busiest <= slack * even:
            return sorted(sharded), bins
        sharded.add(cand[0])


# ---------------------------------------------------------------------------
# run time: one process per GPU (launched by torch.distributed.run / torchrun or anything else that sets RANK /
# WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT).  The plumbing -- rendezvous, barriers, small host objects --
# goes over catch_amd.netstore.TcpGroup (plain sockets; no torch anywhere in the product: the gloo stand-in the CPU
# tests use lives in tests/gloo_group.py since round 6); the data path's exchanges
# are RCCL all-reduces on device buffers (catchhip_shard_allreduce) over a communicator attached to a context of
# its own (a context with a communicator switches catchhip_setcover_greedy to the per-pick sharded form, which
# the whole-group solves must not take).
# ---------------------------------------------------------------------------
class World:
    def __init__(self, rank=0, size=1, group=None, comm_ctx=None, rccl=False):
        self.rank, self.size, self.group, self.comm_ctx = rank, size, group, comm_ctx
        self.rccl = rccl

    @property
    def dist(self):
        """None for a single process (what callers test), else the group."""
        return self.group

    def barrier(self):
        if self.group is not None:
            self.group.barrier()

    def allgather(self, obj):
        """Every rank's host object, in rank order."""
        if self.group is None:
            return [obj]
        return self.group.allgather(obj)

    def broadcast(self, obj, src=0):
        """Rank src's host object on every rank."""
        if self.group is None:
            return obj
        return self.group.broadcast(obj, src)

    def agree(self, error):
        """Collective error check: every rank passes its own failure (a string)
        or None; if any rank failed, EVERY rank raises -- a rank that left a
        collective section alone would leave the others waiting in the next
        all-reduce for ever."""
        errs = self.allgather(error)
        bad = [(r, e) for r, e in enumerate(errs) if e]
        if bad:
            raise RuntimeError("multi-rank filter: " + "; ".join("rank %d: %s" % be for be in bad))

    def exchange_for(self, shards):
        """The exchange callable sharded_solve wants: RCCL on the device buffers,
        or -- CATCHHIP_EXCHANGE=gloo (any value but "rccl"), for boxes where RCCL cannot span the ranks
        (e.g. several ranks on ONE GPU) -- through host memory over the group."""
        if self.rccl:
            def exchange(which):
                for sh in shards:
                    sh.allreduce(which)
            return exchange
        return lambda which: host_exchange(self.group, shards, which)

    def native_for(self, shards):
        """What sharded_solve's `native` should be for these shards: "rccl" when the ranks have a communicator (one
        process per GPU: the whole loop then runs under the C ABI, catchhip_shard_solve), None when the exchange goes
        through the host (several ranks on one GPU, or the CATCHHIP_SHARD_PYTHON_LOOP test hook)."""
        from catch_amd import _lib
        if self.rccl and len(shards) == 1 and not _lib.test_env("CATCHHIP_SHARD_PYTHON_LOOP"):
            return "rccl"
        return None

    def close(self):
        if self.group is not None:
            self.group.close()
            self.group = None


_world = World()


def world():
    """The process's World (single process unless init_from_env ran)."""
    return _world


def init_from_env():
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment (torch.distributed.run sets them): a TCP group
    for the plumbing and an RCCL communicator on a dedicated context of this rank's GPU.  Idempotent; a no-op for
    WORLD_SIZE <= 1."""
    global _world
    import os
    import sys
    size = int(os.environ.get("WORLD_SIZE", "1"))
    if size <= 1 or _world.size > 1:
        return _world
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import ctypes
    from catch_amd import engine, netstore
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # RCCL announces itself on stdout; callers print machine-readable lines
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        group = netstore.TcpGroup(rank, size, os.environ["MASTER_ADDR"], os.environ.get("CATCHHIP_STORE_PORT"))
        group.barrier()
        ctypes.CDLL(None).fflush(None)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    ndev = max(1, engine.device_count())
    device = local_rank % ndev
    os.environ.setdefault("CATCHHIP_DEVICE", str(device))   # engine.default_context()
    uid = group.broadcast(engine.Context.comm_unique_id() if rank == 0 else None, 0)
    comm_ctx = engine.Context(device)
    rccl = os.environ.get("CATCHHIP_EXCHANGE", "rccl") == "rccl"
    if rccl:
        # RCCL refuses e.g. two ranks on one device; the ranks then agree (over the group)
        # to exchange through the host instead of dying one by one
        err = None
        try:
            comm_ctx.comm_init(uid, size, rank)
        except Exception as exc:   # noqa: BLE001 -- whatever the C ABI maps the RCCL error to
            err = "%s: %s" % (type(exc).__name__, exc)
        errs = group.allgather(err)
        if not any(e is not None for e in errs):
            # every rank has a communicator: one checked all-reduce before anything depends on it
            try:
                comm_ctx.comm_selftest(1 << 20)
            except Exception as exc:   # noqa: BLE001
                err = "%s: %s" % (type(exc).__name__, exc)
            errs = group.allgather(err)
        if any(e is not None for e in errs):
            if err is None:
                comm_ctx.comm_destroy()
            if rank == 0:
                print("catch_amd.parallel: RCCL communicator not available (%s); the solver rounds "
                      "exchange through host memory over the process group" % next(e for e in errs if e is not None),
                      file=sys.stderr)
            rccl = False
    _world = World(rank, size, group, comm_ctx, rccl)
    return _world


def host_exchange(group, shards, which):
    """All-reduce of the shards' gain (SUM) or lost (MAX) buffers through host
    memory: over the shards of this process first, then over the process group
    (None: single process; else anything with netstore.TcpGroup's allreduce -- the CPU tests pass tests/gloo_group.py's).
    Fallback transport and the one the CPU tests use."""
    import numpy as np
    bufs = [sh.buffer_to_host(which) for sh in shards]
    acc = bufs[0].astype(np.int64)
    for b in bufs[1:]:
        acc = acc + b if which == 0 else np.maximum(acc, b)
    if group is not None and acc.size:      # (the size is the same on every rank)
        acc = group.allreduce(acc, "sum" if which == 0 else "max")
    for sh in shards:
        sh.buffer_from_host(which, acc)
