// Host side of the connected-components search of the clustering step -- no kernel in this file.
//
// catch/utils/cluster.py:235-355 explores a graph depth first with an early-stop rule (a neighbour within the
// early-stop distance is absorbed without being explored), so the components depend on the ORDER in which an
// explored vertex's neighbours are looked at: the iteration order of the Python set `indices_to_consider -
// indices_to_visit_or_already_visited` (:303-304).  Round 6: the whole search runs here (catchhip_dfs_run_all), with
// the one set whose LAYOUT matters -- indices_to_consider, called `remaining` below -- kept as CPython keeps it
// (PyIntSet: Objects/setobject.c 3.7-3.12 for keys that are small non-negative ints, hash(i) == i; the same
// emulation core.hip's chip_pyset_order has done for the near-duplicate filters' list(set) since round 3).  An
// explored vertex asks for the order of its still-unqueued neighbours only when it has two or more of them, and then
// one of three cases answers (catch_amd/utils/cluster.py _components states them):
//   ascending    the difference's table has more slots than there are vertices: every key in the slot of its own
//                value, iteration by slot
//   copy rank    len(remaining) // 4 > len(queued): CPython builds the difference as remaining.copy() minus the
//                members of queued; the copy's slot order is computed once per component
//   built        otherwise the survivors are inserted one by one into a fresh set, in remaining's slot order, with
//                the table rebuilt as it grows: simulated insert by insert
// Until round 5 the third case and the copy were handed back to the interpreter (real sets, ~100 us each, 0.9 s of
// an S5 step with the GPU idle); the step-wise entry points (catchhip_dfs_run & co.) remain as the cross-check the
// tests run both ways.  catch_amd/utils/cluster.py compares PyIntSet with the interpreter's own sets once per
// process (catchhip_pyintset_*) and falls back to the step-wise path on any difference.
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

// ---- a CPython set of small non-negative ints, slot for slot ------------------------------------------------------
// Objects/setobject.c (3.7-3.12): open addressing over a power-of-two table; the home slot of a key is hash & mask
// (hash(i) == i for 0 <= i < 2^61 - 1), a probe looks at the home slot and, when home + 9 <= mask, the nine slots
// after it, then jumps to (5 i + 1 + (perturb >>= 5)) & mask.  set_add_entry rebuilds the table when fill * 5 >=
// mask * 3 after an insertion (to the smallest power of two above 4 x used, 2 x used beyond 50,000 entries),
// re-inserting the entries in slot order; a discard leaves a dummy (counted in fill, not in used) and moves nothing;
// `a -= b` ends by rebuilding when more than mask / 4 slots are dummies; set_merge into an empty set (copy()) first
// grows the target to the smallest power of two above 2 x used when used * 5 >= mask * 3 of the 8-slot table, then
// copies slot for slot when the two tables have the same size and the source has no dummies, and otherwise
// re-inserts in the source's slot order; iteration is by slot.
struct PyIntSet {
    static constexpr u32 EMPTY = 0xffffffffu, DUMMY = 0xfffffffeu;
    std::vector<u32> tab;
    size_t mask = 7, fill = 0, used = 0;
    void clear8() { tab.assign(8, EMPTY); mask = 7; fill = used = 0; }
    static void insert_clean(u32 *t, size_t mask, u32 key) {
        size_t perturb = key, i = key & mask;
        for (;;) {
            if (t[i] == EMPTY) { t[i] = key; return; }
            if (i + 9 <= mask)
                for (size_t j = 1; j <= 9; ++j)
                    if (t[i + j] == EMPTY) { t[i + j] = key; return; }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    // set_table_resize: entries re-inserted in slot order, dummies dropped
    void resize(size_t minused) {
        size_t nsize = 8;
        while (nsize <= minused) nsize <<= 1;
        std::vector<u32> nt(nsize, EMPTY);
        for (size_t s = 0; s <= mask; ++s)
            if (tab[s] < DUMMY) insert_clean(nt.data(), nsize - 1, tab[s]);
        tab.swap(nt);
        mask = nsize - 1;
        fill = used;
    }
    // set_add_entry for a key that is not in the set, in a table without dummies (fresh sets only)
    void add_new(u32 key) {
        insert_clean(tab.data(), mask, key);
        ++fill; ++used;
        if (fill * 5 >= mask * 3) resize(used > 50000 ? used * 2 : used * 4);
    }
    // slot of a key (set_lookkey), or (size_t)-1
    size_t find(u32 key) const {
        size_t perturb = key, i = key & mask;
        for (;;) {
            if (tab[i] == key) return i;
            if (tab[i] == EMPTY) return (size_t)-1;
            if (i + 9 <= mask)
                for (size_t j = 1; j <= 9; ++j) {
                    if (tab[i + j] == key) return i + j;
                    if (tab[i + j] == EMPTY) return (size_t)-1;
                }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    bool discard(u32 key) {
        const size_t s = find(key);
        if (s == (size_t)-1) return false;
        tab[s] = DUMMY; --used;
        return true;
    }
    // the tail of set_difference_update_internal
    void after_difference_update() {
        if (fill - used > mask / 4) resize(used > 50000 ? used * 2 : used * 4);
    }
    // set(range(n)): n insertions of ascending keys
    void init_range(u32 n) {
        clear8();
        for (u32 k = 0; k < n; ++k) add_new(k);
    }
    // the table of other.copy() (make_new_set -> set_merge into an empty set)
    void copy_of(const PyIntSet &o) {
        clear8();
        if (o.used == 0) return;
        if (o.used * 5 >= mask * 3) {
            size_t nsize = 8;
            while (nsize <= o.used * 2) nsize <<= 1;
            tab.assign(nsize, EMPTY);
            mask = nsize - 1;
        }
        if (mask == o.mask && o.fill == o.used) tab = o.tab;
        else
            for (size_t s = 0; s <= o.mask; ++s)
                if (o.tab[s] < DUMMY) insert_clean(tab.data(), mask, o.tab[s]);
        fill = used = o.used;
    }
    template <class F> void for_each(F f) const {
        for (size_t s = 0; s <= mask; ++s)
            if (tab[s] < DUMMY) f(tab[s]);
    }
};

// test surface (catch_amd/utils/cluster.py checks the emulation against the interpreter's sets once per process)
struct catchhip_pyintset { PyIntSet s; u32 n = 0; std::vector<u32> out; };

extern "C" int catchhip_pyintset_create(u32 n, catchhip_pyintset **out) {
    ARG_CHECK(out && n < PyIntSet::DUMMY);
    catchhip_pyintset *h = new (std::nothrow) catchhip_pyintset();
    if (!h) return CATCHHIP_ENOMEM;
    h->n = n;
    h->s.init_range(n);
    *out = h;
    return 0;
}
extern "C" void catchhip_pyintset_destroy(catchhip_pyintset *h) { delete h; }
// s -= set(keys)   (keys distinct)
extern "C" int catchhip_pyintset_isub(catchhip_pyintset *h, const u32 *keys, i64 count) {
    ARG_CHECK(h && count >= 0 && (count == 0 || keys));
    for (i64 i = 0; i < count; ++i) { ARG_CHECK(keys[i] < h->n); h->s.discard(keys[i]); }
    h->s.after_difference_update();
    return 0;
}
// which: 0 list(s), 1 list(s.copy()), 2 list(s - set(keys)) as CPython builds it (a copy with the members of keys
// discarded when len(s) // 4 > len(keys), the survivors inserted into a fresh set otherwise); keys: distinct members of s
extern "C" int catchhip_pyintset_list(catchhip_pyintset *h, i32 which, const u32 *keys, i64 count, const u32 **p, i64 *n) {
    ARG_CHECK(h && p && n && which >= 0 && which <= 2 && count >= 0 && (count == 0 || keys));
    h->out.clear();
    if (which == 0) h->s.for_each([&](u32 k) { h->out.push_back(k); });
    else {
        std::vector<u8> gone(h->n, 0);
        if (which == 2) for (i64 i = 0; i < count; ++i) { ARG_CHECK(keys[i] < h->n); gone[keys[i]] = 1; }
        PyIntSet t;
        if (which == 1 || (i64)(h->s.used >> 2) > count) {
            t.copy_of(h->s);
            t.for_each([&](u32 k) { if (!gone[k]) h->out.push_back(k); });
        } else {
            t.clear8();
            h->s.for_each([&](u32 k) { if (!gone[k]) t.add_new(k); });
            t.for_each([&](u32 k) { h->out.push_back(k); });
        }
    }
    *p = h->out.data(); *n = (i64)h->out.size();
    return 0;
}

// ---- the whole search --------------------------------------------------------------------------------------------
// comp[n]: the vertices component after component in the order the components are found (a component's vertices in
// the order they entered it), comp_ptr[*ncomp + 1] (room for n + 1).  stats[8]: explored vertices whose difference
// iterates ascending / is a copy of `remaining` / is built insert by insert (the three cases of
// catch_amd/utils/cluster.py, counted as the step-wise path counts them), how many of all those had fewer than two new
// neighbours (no order to establish), copies simulated, differences simulated, keys those inserted, orders read off
// the home slots.
extern "C" int catchhip_dfs_run_all(catchhip_dfs *d, u32 *comp, i64 *comp_ptr, i64 *ncomp, i64 *stats) {
    ARG_CHECK(d && comp && comp_ptr && ncomp && !d->in_component && d->next_start == 0);
    const u32 n = d->n;
    PyIntSet R, T;
    R.init_range(n);
    std::vector<u32> order;              // members of R in slot order, listed when a built difference first needs them
    bool have_order = false;
    std::vector<u32> slot(n, 0);         // rank of a vertex in the copy of R
    std::vector<std::pair<u32, u32>> nb; // (sort key, edge)
    i64 st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    i64 at = 0, nc = 0;
    u32 hi = n - 1;                      // the largest member of R (start is the smallest)
    comp_ptr[0] = 0;
    for (u32 start = 0; start < n; ++start) {
        if (!d->avail[start]) continue;  // in an earlier component
        while (!d->avail[hi]) --hi;      // (start itself is a member)
        const i64 m = (i64)R.used;
        i64 q = 1;
        bool have_rank = false;
        d->stack.clear();
        d->stack.push_back(start);
        d->avail[start] = 0;
        const i64 first = at;
        while (!d->stack.empty()) {
            const u32 j = d->stack.back();
            d->stack.pop_back();
            comp[at++] = j;              // (a stacked vertex is never absorbed: it is queued, so nobody lists it again)
            if (m == q) continue;
            nb.clear();
            for (i64 e = d->ptr[j]; e < d->ptr[j + 1]; ++e)
                if (d->avail[d->idx[e]]) nb.emplace_back(d->idx[e], (u32)e);
            const int which = diff_iterates_ascending(n, m, q) ? 0 : (m >> 2) > q ? 1 : 2;
            st[which] += 1;
            if (nb.size() < 2) st[3] += 1;
            else if (which != 0) {       // (which == 0: the lists are ascending)
                // Slots of the difference's table.  When the members of `remaining` span fewer keys than that, no two
                // of them share a home slot (key & mask): whatever the order of insertion and however often the
                // table was rebuilt on the way, every key sits at home and iteration is by key & mask -- the
                // ascending case is the special case "more slots than vertices".  Components peel the vertices off
                // from the low end, so this is the rule; the simulations below are the exception.
                u64 size;
                if (which == 1) { size = 8; if (m * 5 >= 21) while (size <= 2 * (u64)m) size <<= 1; }
                else size = table_size_after_inserts(m - q);
                if ((u64)(hi - start) < size) {
                    const u32 msk = (u32)(size - 1);
                    for (auto &t : nb) t.first = t.first & msk;
                    st[7] += 1;
                } else if (which == 1) {
                    if (!have_rank) {
                        T.copy_of(R);
                        u32 r = 0;
                        T.for_each([&](u32 k) { slot[k] = r++; });
                        have_rank = true;
                        st[4] += 1;
                    }
                    for (auto &t : nb) t.first = slot[t.first];
                } else {
                    if (!have_order) {
                        order.clear();
                        R.for_each([&](u32 k) { order.push_back(k); });
                        have_order = true;
                    }
                    T.clear8();
                    for (const u32 k : order)
                        if (d->avail[k]) T.add_new(k);
                    st[5] += 1;
                    st[6] += (i64)T.used;
                    for (auto &t : nb) t.first = (u32)T.find(t.first);
                }
                std::sort(nb.begin(), nb.end());
            }
            for (const auto &t : nb) {
                const u32 k = d->idx[t.second];
                d->avail[k] = 0;
                q += 1;
                if (d->com[t.second] >= d->near_common) comp[at++] = k;    // absorbed: in the component, never explored
                else d->stack.push_back(k);
            }
        }
        // indices_to_consider -= cc
        for (i64 i = first; i < at; ++i) R.discard(comp[i]);
        R.after_difference_update();
        have_order = false;
        comp_ptr[++nc] = at;
    }
    d->next_start = n;
    *ncomp = nc;
    if (stats) for (int i = 0; i < 8; ++i) stats[i] = st[i];
    return 0;
}
