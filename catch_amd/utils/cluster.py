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

import numpy as np

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


def _components(n, row_fn, threshold, early_stop_threshold):
    """Connected components by depth-first search with the reference's
    early-stop heuristic (:235-355): a neighbour within early_stop_threshold is
    absorbed into the component without being explored itself, so the result
    can depend on the order neighbours are examined in.  That order is the
    iteration order of `remaining - queued`; the same set operations are
    applied to the same sets in the same sequence here, so CPython produces
    the same order.  row_fn(j, candidates as an int64 array) -> float64 distances."""
    remaining = set(range(n))
    done = set()
    components = []
    for start in range(n):
        if start in done:
            continue
        seen = set()
        stack = [start]
        queued = {start}
        while len(stack) > 0:
            j = stack.pop()
            if j in seen:
                continue
            seen.add(j)
            diff = remaining - queued
            if not diff:
                continue
            # the set's own iteration order, as an index array in one pass
            cand = np.fromiter(diff, dtype=np.int64, count=len(diff))
            d = row_fn(j, cand)
            adjacent = np.nonzero(d <= threshold)[0]
            near = d[adjacent] <= early_stop_threshold
            for k, is_near in zip(cand[adjacent].tolist(), near.tolist()):
                if is_near:
                    seen.add(k)
                else:
                    stack.append(k)
                queued.add(k)
        done.update(seen)
        remaining -= seen
        components.append(sorted(seen))
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
    return _components(sigs.n, row, threshold, early_stop_threshold)


def cluster_with_minhash_signatures(seqs, k=12, N=100, threshold=0.1,
                                    cluster_method="simple"):
    """Clusters of sequence names, largest first (:358-430)."""
    num_seqs = len(seqs)
    names = list(seqs.keys())
    logger.info("Producing signatures of %d sequences", num_seqs)
    family = lsh.MinHashFamily(k, N=N)
    if cluster_method not in ("simple", "hierarchical"):
        family._draw()
        raise ValueError("Unknown cluster_method '%s'" % cluster_method)
    jaccard_dist_threshold = _jaccard_dist_from_mash_dist(threshold, k)
    if num_seqs == 0:
        family._draw()
        return []
    sigs = family.signatures([seqs[n] for n in names])
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
    return [[names[i] for i in c] for c in clusters]
