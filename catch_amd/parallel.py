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
def sharded_solve(shards, exchange):
    """Runs the rounds of one instance over the shards THIS process holds
    (normally one) and returns the picks in the sequential pick order.
    Raises IndexError-like CatchHipError from picks() when the rank list is
    exhausted, as the unsharded solver does."""
    while True:
        for sh in shards:
            sh.count()
        exchange(0)
        for sh in shards:
            sh.claim_check()
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


def plan_with_sharding(group_costs, world, min_cost=0):
    """Two-level plan.  A group whose cost exceeds an even share of the total
    (and min_cost) is SHARDED over all ranks (every rank takes 1/world of it);
    the others go whole to ranks, longest first, onto the least loaded rank.
    Returns (sharded: list of group indices, whole: world lists of indices)."""
    costs = list(group_costs)
    if world <= 1:
        return [], [sorted(range(len(costs)), key=lambda i: (-costs[i], i))]
    share = sum(costs) / float(world)
    sharded = [i for i in range(len(costs))
               if costs[i] > share and costs[i] >= min_cost]
    rest = [i for i in range(len(costs)) if i not in set(sharded)]
    order = sorted(rest, key=lambda i: (-costs[i], i))
    base = sum(costs[i] for i in sharded) / float(world)
    heap = [(base, b) for b in range(world)]
    bins = [[] for _ in range(world)]
    for i in order:
        load, b = heapq.heappop(heap)
        bins[b].append(i)
        heapq.heappush(heap, (load + costs[i], b))
    return sharded, bins
