"""Clustering of sequences before design (mirrors catch/utils/cluster.py; used
by `--cluster-and-design-separately`).

On the device (catch_amd/csrc/cluster.hip): the MinHash signatures of all
sequences (md5 of every k-mer, N smallest per sequence) and the signature
distances -- one row at a time for the connected-components search, or the
whole condensed matrix for hierarchical clustering.  On the host: the
depth-first search itself (sequential by nature: which vertices get explored
depends on what earlier explorations absorbed, cluster.py:235-355) and
SciPy's average linkage (:197-232), the same third-party routine the
reference calls.
"""
from collections import defaultdict
import logging
import os

import numpy as np

from catch_amd import _lib
from catch_amd.utils import lsh

logger = logging.getLogger(__name__)


def _jaccard_dist_from_mash_dist(mash_dist, k):
    """Jaccard distance whose Mash distance is `mash_dist` (Ondov et al. 2016,
    eq. 4 solved for j; cluster.py:47-68)."""
    return 1.0 - 1.0 / (2.0 * np.exp(k * mash_dist) - 1)


def set_max_num_processes_for_computing_distances(max_num_processes=8):
    """Accepted for compatibility (cluster.py:71-84); distances are computed
    on the GPU."""


def make_signatures_with_minhash(family, seqs):
    """dict name -> signature tuple, one hash function for all (:28-44)."""
    names = list(seqs.keys())
    sigs = family.signatures([seqs[n] for n in names])
    try:
        arr = sigs.fetch()
    finally:
        sigs.close()
    return {n: tuple(int(x) for x in arr[i]) for i, n in enumerate(names)}


def create_condensed_dist_matrix(n, dist_fn, num_processes=None):
    """Condensed float32 distance matrix of an arbitrary Python distance
    function (:102-194).  (Signature distances do not come through here: see
    engine.Signatures.condensed.)"""
    out = np.zeros(n * (n - 1) // 2, dtype=np.float32)
    at = 0
    for i in range(n):
        for j in range(i + 1, n):
            out[at] = dist_fn(i, j)
            at += 1
    return out


def cluster_hierarchically_from_dist_matrix(dist_matrix, threshold):
    """Average-linkage clusters cut at `threshold`, largest first (:197-232)."""
    from scipy.cluster import hierarchy
    if len(dist_matrix) == 0:
        return [[0]]
    linkage = hierarchy.linkage(dist_matrix, method="average")
    labels = hierarchy.fcluster(linkage, threshold, criterion="distance")
    members = defaultdict(list)
    for i, c in enumerate(labels):
        members[int(c)].append(i)
    numbers = list(range(min(members), max(members) + 1))
    numbers.sort(key=lambda c: len(members[c]), reverse=True)   # stable
    return [members[c] for c in numbers]


def _table_size_after_inserts(m):
    """Slots of a CPython set's hash table after m insertions into an empty
    set (setobject.c, 3.7 and later: a table starts with 8 slots and is rebuilt
    when fill * 5 >= mask * 3, to the smallest power of two above used * 4, or
    above used * 2 beyond 50,000 entries)."""
    size, used = 8, 0
    while True:
        mask = size - 1
        first = -(-3 * mask // 5)             # the insertion that triggers the rebuild
        if m < first:
            return size
        used = first
        want = used * (2 if used > 50000 else 4)
        size = 8
        while size <= want:
            size <<= 1
        if m == used:
            return size


def _diff_iterates_ascending(n, m, q):
    """True when `remaining - queued` (|remaining| = m, queued a subset of it
    with q elements, all elements in range(n)) is certain to iterate in
    ascending order: its hash table then has more than n - 1 slots, every int
    sits in the slot of its own value (hash(i) == i) and iteration is by slot.
    CPython builds the difference in one of two ways (setobject.c
    set_difference): when len(remaining) // 4 > len(queued) it copies
    `remaining` into a table of the smallest power of two above 2 m slots and
    discards the members of `queued` (never enough of them to trigger a
    rebuild); otherwise it inserts the m - q survivors one by one into a fresh
    set.  Any other table may hold displaced entries, and its order is left to
    the interpreter (the caller then builds the real set)."""
    if (m >> 2) > q:
        size = 8
        if m * 5 >= 21:
            while size <= 2 * m:
                size <<= 1
    else:
        size = _table_size_after_inserts(m - q)
    return size > n - 1


_fast_order_ok = None


def _fast_order_available():
    """Whether this interpreter's sets behave as _diff_iterates_ascending and
    the copy shortcut of _components assume (checked once on a few hundred
    differences; any surprise disables the shortcuts and every difference is
    built for real)."""
    global _fast_order_ok
    if _fast_order_ok is None:
        import random as _random
        rnd = _random.Random(12345)      # (a private generator: the callers' `random` stream is untouched)
        ok = True
        for n in (9, 40, 300, 2500, 70000):
            remaining = set(range(n))
            for _ in range(40):
                m = len(remaining)
                if m == 0:
                    break
                pool = rnd.sample(sorted(remaining), min(m, rnd.choice((1, 3, m // 5 + 1, m // 3 + 1))))
                queued = set()
                for k in pool:
                    queued.add(k)
                if _diff_iterates_ascending(n, m, len(queued)):
                    d = list(remaining - queued)
                    ok = ok and d == sorted(d)
                if (m >> 2) > len(queued):
                    # the difference is a copy of `remaining` with the members of `queued` taken out
                    ok = ok and list(remaining - queued) == [x for x in remaining.copy() if x not in queued]
                remaining -= set(pool[:len(pool) // 2 + 1])
        _fast_order_ok = ok
    return _fast_order_ok


_path_counts = {"ascending": 0, "copy rank": 0, "real difference": 0, "list calls": 0, "graphs": 0}   # (tests look at these)


def _components(n, row_fn, threshold, early_stop_threshold, neighbors_fn=None, neighbors_many_fn=None,
                batch=32, local_lists=False):
    """Connected components by depth-first search with the reference's
    early-stop heuristic (:235-355): a neighbour within early_stop_threshold is
    absorbed into the component without being explored itself, so the result
    can depend on the order neighbours are examined in.  That order is the
    iteration order of `remaining - queued`; the same set operations are
    applied to the same sets in the same sequence here, so CPython produces
    the same order.  row_fn(j, candidates as an int64 array) -> float64 distances.
    neighbors_fn(j) -> (indices ascending, distances) of ALL vertices within
    `threshold` of j, or None.  With it the O(n) set difference and distance
    row per explored vertex are avoided in two situations:
      * the difference is known to iterate in ascending order
        (_diff_iterates_ascending -- about the first two thirds of a search):
        the neighbours that count are those still in `remaining` and not yet
        queued, in ascending order;
      * CPython builds the difference as a COPY of `remaining` minus the
        members of `queued` (len(remaining) // 4 > len(queued); the discards do
        not move anything and are too few to trigger a rebuild): the order is
        that of `remaining.copy()`, which changes only when `remaining` does --
        once per component -- so one real copy per component ranks every
        vertex and the neighbours are sorted by that rank.
    Otherwise the difference is built for real.
    neighbors_many_fn(list of vertices) -> list of such pairs: the lists of the
    vertex being explored and of the vertices on top of the stack (the next to
    be explored) in one device call.  local_lists: neighbors_fn slices a graph
    that is already on the host (one device pass made all the lists)."""
    remaining = set(range(n))
    done = set()
    components = []
    lists = neighbors_fn is not None and _fast_order_available()
    if lists:
        in_remaining = np.ones(n, dtype=bool)
        queued_in = np.zeros(n, dtype=np.int64)      # the component (1-based) that queued the vertex
    cache = {}
    comp_no = 0
    for start in range(n):
        if start in done:
            continue
        comp_no += 1
        seen = set()
        stack = [start]
        queued = {start}
        copy_rank = None           # rank of every vertex in the iteration order of remaining.copy()
        cache.clear()
        if lists:
            queued_in[start] = comp_no
        while len(stack) > 0:
            j = stack.pop()
            if j in seen:
                continue
            seen.add(j)
            # (queued is a subset of remaining: its members come out of differences with it)
            m, q = len(remaining), len(queued)
            if m == q:
                continue
            ascending = lists and _diff_iterates_ascending(n, m, q)
            if ascending or (lists and (m >> 2) > q):
                hit = cache.pop(j, None)
                if hit is None:
                    if not local_lists:                      # (local: slices of a graph held on the host)
                        _path_counts["list calls"] += 1
                    if neighbors_many_fn is None:
                        hit = neighbors_fn(j)
                    else:
                        ask = [j]
                        for k in reversed(stack[-4 * batch:]):       # (a vertex is stacked at most once)
                            if len(ask) >= batch:
                                break
                            if k not in seen and k not in cache:
                                ask.append(k)
                        got = neighbors_many_fn(ask)
                        for k, r in zip(ask[1:], got[1:]):
                            cache[k] = r
                        hit = got[0]
                idx, dist = hit
                keep = in_remaining[idx] & (queued_in[idx] != comp_no)
                ks, dk = idx[keep], dist[keep]
                _path_counts["ascending" if ascending else "copy rank"] += 1
                if not ascending and len(ks) > 1:
                    if copy_rank is None:
                        cp = remaining.copy()
                        members = np.fromiter(cp, dtype=np.int64, count=len(cp))
                        copy_rank = np.empty(n, dtype=np.int64)
                        copy_rank[members] = np.arange(members.size)
                    o = np.argsort(copy_rank[ks], kind="stable")
                    ks, dk = ks[o], dk[o]
                near = dk <= early_stop_threshold
            else:
                _path_counts["real difference"] += 1
                diff = remaining - queued
                if not diff:
                    continue
                # the set's own iteration order, as an index array in one pass
                cand = np.fromiter(diff, dtype=np.int64, count=len(diff))
                d = row_fn(j, cand)
                adjacent = np.nonzero(d <= threshold)[0]
                ks, near = cand[adjacent], d[adjacent] <= early_stop_threshold
            if len(ks):
                # near ones are absorbed, the others explored later, in this order; all are queued
                seen.update(ks[near].tolist())
                stack.extend(ks[~near].tolist())
                queued.update(ks.tolist())
                if lists:
                    queued_in[ks] = comp_no
        done.update(seen)
        remaining -= seen
        if lists:
            in_remaining[np.fromiter(seen, dtype=np.int64, count=len(seen))] = False
        components.append(sorted(seen))
    components.sort(key=len, reverse=True)
    return components


last_timings = {}        # wall seconds of the last cluster_with_minhash_signatures call by stage (tools, bench.py)
_native_sets_ok = None
_native_stats = {}      # catchhip_dfs_run_all's extra counters, summed (tools/s5_profile.py prints them)


def _native_sets_available():
    """Whether the library's emulation of a CPython set of small ints (catch_amd/csrc/components.hip, PyIntSet)
    lays its tables out as THIS interpreter does: set(range(n)), `-=`, copy() and both forms of `a - b` are
    compared with real sets on ~150 cases, once per process (~0.1 s).  Any difference sends the
    search back to the step-wise path, where every set whose layout matters is a real one."""
    global _native_sets_ok
    if _native_sets_ok is None:
        import ctypes
        L = _lib.lib()
        rs = np.random.RandomState(4242)     # (a private generator: the callers' random streams are untouched)
        p32, cnt = _lib.c_u32p(), ctypes.c_int64(0)

        def listed(h, which, keys=()):
            k = np.ascontiguousarray(np.fromiter(keys, dtype=np.uint32, count=len(keys)))
            _lib.check(L.catchhip_pyintset_list(h, which, k.ctypes.data_as(_lib.c_u32p), int(k.size),
                                                ctypes.byref(p32), ctypes.byref(cnt)))
            return np.ctypeslib.as_array(p32, shape=(cnt.value,)).tolist() if cnt.value else []
        ok = True
        # (60,000: beyond 50,000 entries a rebuilt table is 2 x, not 4 x, the entries)
        for n, rounds in ((1, 1), (9, 6), (40, 8), (300, 10), (2500, 10), (60000, 2)):
            if not ok:
                break
            h = ctypes.c_void_p()
            _lib.check(L.catchhip_pyintset_create(n, ctypes.byref(h)))
            try:
                remaining = set(range(n))
                for _ in range(rounds):
                    m = len(remaining)
                    if m == 0:
                        break
                    members = np.fromiter(remaining, dtype=np.int64, count=m)
                    ok = ok and listed(h, 0) == members.tolist() and listed(h, 1) == list(remaining.copy())
                    for frac in (0.02, 0.3, 0.7):
                        queued = set()
                        for k in members[rs.permutation(m)[:max(1, min(m, int(m * frac)))]].tolist():
                            queued.add(k)
                        ok = ok and listed(h, 2, queued) == list(remaining - queued)
                    if rs.random_sample() < 0.5:
                        a = int(members[rs.randint(m)])
                        w = max(1, m // int(rs.choice((3, 7, 20))))
                        cc = set(members[(members >= a) & (members < a + w)].tolist())
                    else:
                        cc = set(members[rs.permutation(m)[:max(1, m // int(rs.choice((2, 5, 11))))]].tolist())
                    remaining -= cc
                    k = np.fromiter(cc, dtype=np.uint32, count=len(cc))
                    _lib.check(L.catchhip_pyintset_isub(h, k.ctypes.data_as(_lib.c_u32p), int(k.size)))
            finally:
                L.catchhip_pyintset_destroy(h)
        _native_sets_ok = bool(ok)
    return _native_sets_ok


def _components_over_graph(n, ptr, gidx, gcom, near_common, row_fn, threshold, early_stop_threshold):
    """_components with the neighbour lists of every vertex on the host (CSR), run natively
    (catch_amd/csrc/components.hip).  Round 6: the whole search in one call (catchhip_dfs_run_all) with
    `remaining` emulated slot for slot, when the emulation agrees with this interpreter's sets
    (_native_sets_available); otherwise, and under the CATCHHIP_CLUSTER_STEPWISE test hook, step by step
    (catchhip_dfs_run): the explored vertices whose neighbour order is known without a set difference natively, the
    same real `remaining` set -- whose layout decides which case an explored vertex is -- and the real differences
    and copies built here exactly as _components builds them."""
    import ctypes
    L = _lib.lib()
    h = ctypes.c_void_p()
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    gidx = np.ascontiguousarray(gidx, dtype=np.uint32)
    gcom = np.ascontiguousarray(gcom, dtype=np.uint32)
    _lib.check(L.catchhip_dfs_create(n, ptr.ctypes.data_as(_lib.c_i64p), gidx.ctypes.data_as(_lib.c_u32p),
                                     gcom.ctypes.data_as(_lib.c_u32p), int(near_common), ctypes.byref(h)))
    if not _lib.test_env("CATCHHIP_CLUSTER_STEPWISE") and _native_sets_available():
        try:
            comp = np.empty(n, dtype=np.uint32)
            cptr = np.empty(n + 1, dtype=np.int64)
            ncomp = ctypes.c_int64(0)
            stats = (ctypes.c_int64 * 8)()
            _lib.check(L.catchhip_dfs_run_all(h, comp.ctypes.data_as(_lib.c_u32p), cptr.ctypes.data_as(_lib.c_i64p),
                                              ctypes.byref(ncomp), stats))
        finally:
            L.catchhip_dfs_destroy(h)
        for k, v in zip(("ascending", "copy rank", "real difference"), stats):
            _path_counts[k] += int(v)
        for k, v in zip(("order not needed", "copies", "differences built", "keys inserted", "home-slot orders"), list(stats)[3:]):
            _native_stats[k] = _native_stats.get(k, 0) + int(v)
        _native_stats["searches"] = _native_stats.get("searches", 0) + 1
        cptr = cptr[:ncomp.value + 1]
        components = [np.sort(comp[a:b]).tolist() for a, b in zip(cptr[:-1].tolist(), cptr[1:].tolist())]
        components.sort(key=len, reverse=True)
        return components
    remaining = set(range(n))
    queued = set()
    components = []
    status, vertex, cnt = ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int64(0)
    p32 = _lib.c_u32p()

    def listed(fn):
        _lib.check(fn(h, ctypes.byref(p32), ctypes.byref(cnt)))
        return np.ctypeslib.as_array(p32, shape=(cnt.value,)).copy() if cnt.value else np.zeros(0, dtype=np.uint32)
    try:
        while True:
            _lib.check(L.catchhip_dfs_run(h, len(remaining), ctypes.byref(status), ctypes.byref(vertex)))
            if status.value == 0:
                break
            if status.value == 1:
                seen = listed(L.catchhip_dfs_seen)
                remaining -= set(seen.tolist())
                seen.sort()
                components.append(seen.tolist())
                queued = set()
            elif status.value == 2:
                cp = remaining.copy()
                members = np.fromiter(cp, dtype=np.int64, count=len(cp))
                _lib.check(L.catchhip_dfs_set_copy_members(h, members.ctypes.data_as(_lib.c_i64p), int(members.size)))
            else:
                j = int(vertex.value)
                queued.update(listed(L.catchhip_dfs_new_queued).tolist())
                diff = remaining - queued
                if not diff:
                    continue
                cand = np.fromiter(diff, dtype=np.int64, count=len(diff))
                d = row_fn(j, cand)
                adjacent = np.nonzero(d <= threshold)[0]
                ks = np.ascontiguousarray(cand[adjacent])
                near = np.ascontiguousarray(d[adjacent] <= early_stop_threshold, dtype=np.uint8)
                _lib.check(L.catchhip_dfs_push(h, ks.ctypes.data_as(_lib.c_i64p), near.ctypes.data_as(_lib.c_u8p), int(ks.size)))
        got = (ctypes.c_int64 * 3)()
        _lib.check(L.catchhip_dfs_counts(h, got))
        for k, v in zip(("ascending", "copy rank", "real difference"), got):
            _path_counts[k] += int(v)
    finally:
        L.catchhip_dfs_destroy(h)
    components.sort(key=len, reverse=True)
    return components


def find_connected_components(n, dist_fn, threshold,
                              early_stop_threshold=_jaccard_dist_from_mash_dist(0.02, 12)):
    """Components under an arbitrary Python distance function (:235-355)."""
    def row(j, cand):
        return np.asarray([dist_fn(j, k) for k in cand.tolist()], dtype=np.float64)
    return _components(n, row, threshold, early_stop_threshold)


def _components_of_signatures(sigs, threshold,
                              early_stop_threshold=_jaccard_dist_from_mash_dist(0.02, 12)):
    N = float(sigs.N)

    def row(j, cand):
        common = sigs.common_row(j)
        # float(intersect_count) / union_count, 1.0 - similarity (lsh.py:212-215)
        return 1.0 - common[cand].astype(np.float64) / N

    # the smallest number of shared values whose distance is within the threshold (the distance falls
    # as the number grows; evaluated with the expression above, so the comparison is the same)
    within = np.nonzero(1.0 - np.arange(sigs.N + 1, dtype=np.float64) / N <= threshold)[0]
    neighbors = neighbors_many = None
    if sigs.N <= 112 and len(within) and not _lib.test_env("CATCHHIP_CLUSTER_ROWS_ONLY") \
            and not _lib.test_env("CATCHHIP_CLUSTER_NO_GRAPH") and not _lib.test_env("CATCHHIP_CLUSTER_ONE_BY_ONE") \
            and hasattr(sigs, "graph"):
        # the whole neighbour graph in one device pass (round 4): the search reads its lists from the CSR
        # copy, and the distance rows of the real differences come from it too
        import time as _time
        t0 = _time.perf_counter()
        g = sigs.graph(int(within[0]))
        last_timings["graph_s"] = _time.perf_counter() - t0
        if g is not None:
            ptr, gidx, gcom = g
            lut = 1.0 - np.arange(sigs.N + 1, dtype=np.float64) / N
            far = np.full(sigs.n, 2.0, dtype=np.float64)        # scratch: 2.0 = not a neighbour

            def neighbors_g(j):
                a, b = ptr[j], ptr[j + 1]
                return gidx[a:b], lut[gcom[a:b]]

            def row_g(j, cand):
                a, b = ptr[j], ptr[j + 1]
                nb = gidx[a:b]
                far[nb] = lut[gcom[a:b]]
                d = far[cand]
                far[nb] = 2.0
                return d
            _path_counts["graphs"] += 1
            if _fast_order_available() and not _lib.test_env("CATCHHIP_CLUSTER_PYTHON_SEARCH"):
                # common counts whose distance is within the early-stop threshold (same expression, same comparison)
                near = np.nonzero(lut <= early_stop_threshold)[0]
                near_common = int(near[0]) if len(near) else sigs.N + 1
                return _components_over_graph(sigs.n, ptr, gidx, gcom, near_common, row_g, threshold, early_stop_threshold)
            return _components(sigs.n, row_g, threshold, early_stop_threshold, neighbors_g, None, local_lists=True)
    if sigs.N <= 176 and len(within) and not _lib.test_env("CATCHHIP_CLUSTER_ROWS_ONLY"):
        min_common = int(within[0])

        def neighbors(j):
            idx, common = sigs.neighbors(j, min_common)
            return idx, 1.0 - common.astype(np.float64) / N
        if sigs.N <= 112 and not _lib.test_env("CATCHHIP_CLUSTER_ONE_BY_ONE"):
            def neighbors_many(js):
                return [(idx, 1.0 - common.astype(np.float64) / N) for idx, common in sigs.neighbors_many(js, min_common)]
    return _components(sigs.n, row, threshold, early_stop_threshold, neighbors, neighbors_many,
                       batch=getattr(sigs, "NEIGHBORS_MANY_MAX", 32))


def cluster_with_minhash_signatures(seqs, k=12, N=100, threshold=0.1,
                                    cluster_method="simple"):
    """Clusters of sequence names, largest first (:358-430)."""
    num_seqs = len(seqs)
    from catch_amd import engine
    table = seqs if isinstance(seqs, engine.FragmentTable) else None     # (fragments as views: named 0 .. n - 1)
    names = range(num_seqs) if table is not None else list(seqs.keys())
    logger.info("Producing signatures of %d sequences", num_seqs)
    family = lsh.MinHashFamily(k, N=N)
    if cluster_method not in ("simple", "hierarchical"):
        family._draw()
        raise ValueError("Unknown cluster_method '%s'" % cluster_method)
    jaccard_dist_threshold = _jaccard_dist_from_mash_dist(threshold, k)
    if num_seqs == 0:
        family._draw()
        return []
    import time as _time
    t0 = _time.perf_counter()
    last_timings.clear()
    sigs = family.signatures(table if table is not None else [seqs[n] for n in names])
    last_timings["signatures_s"] = _time.perf_counter() - t0
    try:
        if cluster_method == "simple":
            logger.info(("Clustering %d sequences at Jaccard distance "
                         "threshold of %f based on connected components"),
                        num_seqs, jaccard_dist_threshold)
            clusters = _components_of_signatures(sigs, jaccard_dist_threshold)
        else:
            logger.info(("Clustering %d sequences at Jaccard distance "
                         "threshold of %f using hierarchical method"),
                        num_seqs, jaccard_dist_threshold)
            # what the reference's c_float matrix holds: float32(1.0 - c / N)
            lut = (1.0 - np.arange(N + 1, dtype=np.float64) / float(N)).astype(np.float32)
            clusters = cluster_hierarchically_from_dist_matrix(
                sigs.condensed(lut), jaccard_dist_threshold)
    finally:
        sigs.close()
    last_timings["total_s"] = _time.perf_counter() - t0
    last_timings["search_s"] = (last_timings["total_s"] - last_timings["signatures_s"]
                                - last_timings.get("graph_s", 0.0))
    if table is not None:
        return clusters
    return [[names[i] for i in c] for c in clusters]
