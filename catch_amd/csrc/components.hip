// Host side of the connected-components search of the clustering step -- no kernel in this file.
//
// catch/utils/cluster.py:235-355 explores a graph depth first with an early-stop rule (a neighbour within the
// early-stop distance is absorbed without being explored), so the components depend on the ORDER in which an
// explored vertex's neighbours are looked at: the iteration order of the Python set `remaining - queued`.
// catch_amd/utils/cluster.py (_components) establishes when that order is known without building the set:
// ascending while the difference's hash table has more slots than vertices, and the order of `remaining.copy()`
// while CPython builds the difference as a copy.  Those two cases are 96 % of the 224 k explored vertices of
// S5 x 1.0, and each was ~8 us of interpreter; this file runs them natively over the neighbour graph
// (catchhip_sigs_graph) and hands the rest back: a real set difference (status 3), the ranks of
// `remaining.copy()` once per component (status 2), the end of a component (status 1: the caller updates its real
// `remaining` set, whose layout decides all of the above).  No CPython internals are re-implemented here: every
// set whose layout matters stays a real set in the caller.
#include <algorithm>
#include <vector>

#include "internal.h"

struct catchhip_dfs {
    u32 n = 0;
    const i64 *ptr = nullptr;        // borrowed: the caller keeps the graph alive
    const u32 *idx = nullptr, *com = nullptr;
    u32 near_common = 0;             // a neighbour with at least this many common values is absorbed
    std::vector<u8> avail;           // in `remaining` and not queued
    std::vector<u32> stack, seen, queued;
    std::vector<i64> rank;           // of every vertex in remaining.copy(), when have_rank
    bool have_rank = false, in_component = false;
    u32 next_start = 0;
    i64 m = 0, q = 0;                // len(remaining), len(queued)
    size_t queued_read = 0;          // entries of `queued` the caller has fetched
    i64 pending = -1;                // a popped vertex waiting for the ranks
    i64 counts[3] = {0, 0, 0};       // explored with ascending order, by copy rank, handed back
    std::vector<std::pair<i64, u32>> tmp;   // (rank or index, edge)
};

// Slots of a CPython set's table after m insertions into an empty set (catch_amd/utils/cluster.py
// _table_size_after_inserts, which cites setobject.c)
static u64 table_size_after_inserts(i64 mm) {
    u64 size = 8;
    for (;;) {
        const u64 mask = size - 1;
        const i64 first = (i64)((3 * mask + 4) / 5);         // the insertion that triggers the rebuild
        if (mm < first) return size;
        const u64 used = (u64)first;
        const u64 want = used * (used > 50000 ? 2 : 4);
        size = 8;
        while (size <= want) size <<= 1;
        if (mm == first) return size;
    }
}

// catch_amd/utils/cluster.py _diff_iterates_ascending
static bool diff_iterates_ascending(u64 n, i64 m, i64 q) {
    u64 size;
    if ((m >> 2) > q) {
        size = 8;
        if (m * 5 >= 21) while (size <= 2 * (u64)m) size <<= 1;
    } else size = table_size_after_inserts(m - q);
    return size > n - 1;
}

extern "C" int catchhip_dfs_create(u32 n, const i64 *ptr, const u32 *idx, const u32 *com, u32 near_common,
                                   catchhip_dfs **out) {
    ARG_CHECK(out && ptr && n >= 1 && (ptr[n] == 0 || (idx && com)));
    catchhip_dfs *d = new (std::nothrow) catchhip_dfs();
    if (!d) return CATCHHIP_ENOMEM;
    d->n = n; d->ptr = ptr; d->idx = idx; d->com = com; d->near_common = near_common;
    d->avail.assign(n, 1);
    *out = d;
    return 0;
}

extern "C" void catchhip_dfs_destroy(catchhip_dfs *d) { delete d; }

static void dfs_queue(catchhip_dfs *d, u32 k, bool near) {
    d->avail[k] = 0;
    d->q += 1;
    d->queued.push_back(k);
    if (near) d->seen.push_back(k);
}

// Runs until the caller is needed.  m_now = len(remaining) (read when a component starts).
// *status: 0 all vertices are in components, 1 a component has ended (catchhip_dfs_seen has its vertices; take them
// out of `remaining`, then run again), 2 the ranks of remaining.copy() are needed (catchhip_dfs_set_copy_rank, run
// again), 3 the neighbours of *vertex need a real `remaining - queued` (catchhip_dfs_new_queued, catchhip_dfs_push,
// run again).
extern "C" int catchhip_dfs_run(catchhip_dfs *d, i64 m_now, i32 *status, i64 *vertex) {
    ARG_CHECK(d && status && vertex);
    for (;;) {
        if (!d->in_component) {
            while (d->next_start < d->n && !d->avail[d->next_start]) ++d->next_start;
            if (d->next_start >= d->n) { *status = 0; return 0; }
            const u32 start = d->next_start;
            d->in_component = true;
            d->have_rank = false;
            d->m = m_now;
            d->q = 0;
            d->stack.clear(); d->seen.clear(); d->queued.clear();
            d->queued_read = 0;
            d->stack.push_back(start);
            d->avail[start] = 0; d->q = 1; d->queued.push_back(start);
        }
        while (!d->stack.empty() || d->pending >= 0) {
            u32 j;
            if (d->pending >= 0) { j = (u32)d->pending; d->pending = -1; }
            else {
                j = d->stack.back(); d->stack.pop_back();
                d->seen.push_back(j);             // (a stacked vertex is never absorbed: it is queued, so nobody lists it again)
            }
            if (d->m == d->q) continue;
            const bool ascending = diff_iterates_ascending(d->n, d->m, d->q);
            if (!ascending && !((d->m >> 2) > d->q)) {
                d->counts[2] += 1;
                *status = 3; *vertex = j;
                return 0;
            }
            d->tmp.clear();
            for (i64 e = d->ptr[j]; e < d->ptr[j + 1]; ++e)
                if (d->avail[d->idx[e]]) d->tmp.emplace_back((i64)d->idx[e], (u32)e);
            if (!ascending && d->tmp.size() > 1) {
                if (!d->have_rank) { d->pending = j; *status = 2; *vertex = j; return 0; }
                for (auto &t : d->tmp) t.first = d->rank[d->idx[t.second]];
                std::sort(d->tmp.begin(), d->tmp.end());
            }
            d->counts[ascending ? 0 : 1] += 1;
            const size_t s0 = d->stack.size();
            for (const auto &t : d->tmp) {
                const u32 k = d->idx[t.second];
                const bool near = d->com[t.second] >= d->near_common;
                dfs_queue(d, k, near);
                if (!near) d->stack.push_back(k);
            }
            (void)s0;
        }
        d->in_component = false;
        *status = 1;
        return 0;
    }
}

extern "C" int catchhip_dfs_seen(catchhip_dfs *d, const u32 **p, i64 *count) {
    ARG_CHECK(d && p && count);
    *p = d->seen.data(); *count = (i64)d->seen.size();
    return 0;
}

// the vertices queued in this component since the last call (valid until the next call into d)
extern "C" int catchhip_dfs_new_queued(catchhip_dfs *d, const u32 **p, i64 *count) {
    ARG_CHECK(d && p && count);
    *p = d->queued.data() + d->queued_read; *count = (i64)(d->queued.size() - d->queued_read);
    d->queued_read = d->queued.size();
    return 0;
}

extern "C" int catchhip_dfs_set_copy_rank(catchhip_dfs *d, const i64 *rank) {
    ARG_CHECK(d && rank);
    d->rank.assign(rank, rank + d->n);
    d->have_rank = true;
    return 0;
}

// the same from the copy's members in its iteration order (rank = position): only the members' ranks are written -- the
// search only ever asks for vertices that are still in `remaining` -- instead of an n-entry array zeroed, scattered and
// copied per call (444 calls for the 224 k fragments of S5)
extern "C" int catchhip_dfs_set_copy_members(catchhip_dfs *d, const i64 *members, i64 count) {
    ARG_CHECK(d && count >= 0 && (count == 0 || members));
    if (d->rank.size() != d->n) d->rank.assign(d->n, 0);
    for (i64 k = 0; k < count; ++k) {
        ARG_CHECK(members[k] >= 0 && members[k] < (i64)d->n);
        d->rank[(size_t)members[k]] = k;
    }
    d->have_rank = true;
    return 0;
}

// what a real difference found for the vertex of status 3: its neighbours in the difference's order, near[i] != 0
// for the absorbed ones
extern "C" int catchhip_dfs_push(catchhip_dfs *d, const i64 *ks, const u8 *near, i64 count) {
    ARG_CHECK(d && d->in_component && count >= 0 && (count == 0 || (ks && near)));
    for (i64 i = 0; i < count; ++i) ARG_CHECK(ks[i] >= 0 && ks[i] < (i64)d->n && d->avail[ks[i]]);
    for (i64 i = 0; i < count; ++i) {
        dfs_queue(d, (u32)ks[i], near[i] != 0);
        if (!near[i]) d->stack.push_back((u32)ks[i]);
    }
    return 0;
}

extern "C" int catchhip_dfs_counts(const catchhip_dfs *d, i64 *out3) {
    ARG_CHECK(d && out3);
    for (int i = 0; i < 3; ++i) out3[i] = d->counts[i];
    return 0;
}
