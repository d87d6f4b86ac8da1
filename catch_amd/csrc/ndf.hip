// K3: near-duplicate filter with the Hamming-distance LSH family
// (catch/filter/near_duplicate_filter.py:47-142, catch/utils/lsh.py:16-45,
// :218-320).
//
// The reference walks the unique probes in priority order (multiplicity
// descending, stable): a probe that is not yet excluded is kept and excludes
// every not-yet-kept probe that shares a bucket with it in some table and lies
// within Hamming distance dist_thres.  That is the lexicographically-first
// maximal independent set of the graph "share a bucket AND Hamming <= d" in
// priority order.  On the device:
//   1. per table: 64-bit hash of the k sampled characters of every probe,
//      stable radix sort of (hash, index) -> buckets are runs of equal keys,
//      and inside a run indices ascend (= priority descends);
//   2. one lane per sorted slot walks left over its run: every mate with a
//      smaller index is a candidate higher-priority neighbour; verify the
//      Hamming distance on the raw bytes and the sampled characters (exact
//      bucket equality, the hash only groups) and append the edge (i, j);
//   3. rounds over the edge list until every probe is decided: drop a probe
//      with a kept higher-priority neighbour, keep a probe whose
//      higher-priority neighbours are all dropped.
#include <algorithm>
#include <chrono>

#include "internal.h"

// The edge list is written through ONE counter per shard: a single counter is a
// single address (~10 ns per atomic, however the compiler aggregates them), and
// the walks below append edges from inside their loops, once per iteration in
// which some lane found one.  Shard s owns e_i/e_j[s * segcap ..) and the
// counter count[s * ES_STRIDE] (128 bytes apart); wavefronts pick their shard
// round-robin.  count[ES_SHARDS * ES_STRIDE] counts the undecided nodes of a round.
#define ES_SHARDS 64
#define ES_STRIDE 32
#define ES_WORDS (ES_SHARDS * ES_STRIDE + 1)

__global__ void __launch_bounds__(256)
ndf_key_kernel(const u8 *__restrict__ bytes, u32 n, int L, const i32 *__restrict__ pos, int k,
               u64 *__restrict__ keys, u32 *__restrict__ vals, const u32 *__restrict__ grp, size_t pos_group_stride,
               int key_shift = 0) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u8 *p = bytes + (size_t)i * L;
    u64 h = 0xcbf29ce484222325ull;
    if (grp) {   // independent groups: own sampled positions, and the group is part of the key
        pos += (size_t)grp[i] * pos_group_stride;
        h = (h ^ (u64)grp[i]) * 0x100000001b3ull;
    }
    for (int j = 0; j < k; ++j) h = (h ^ (u64)p[pos[j]]) * 0x100000001b3ull;
    keys[i] = h >> key_shift;     // (the key only groups: the lazy resolution sorts its upper 32 bits, half the radix passes)
    vals[i] = i;
}

// rows padded to a multiple of 8 bytes (zeros): the Hamming test reads 8
// characters per load instead of one (byte loads made the edge kernel 76 % of
// the device time of a config-3 design: 20.6 ms per table for 1.3 M probes)
__global__ void __launch_bounds__(256)
ndf_pad_kernel(const u8 *__restrict__ bytes, u32 n, int L, int Lp, u8 *__restrict__ out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (u64)n * Lp) return;
    const u32 i = (u32)(t / Lp), j = (u32)(t - (u64)i * Lp);
    out[t] = j < (u32)L ? bytes[(size_t)i * L + j] : (u8)0;
}

// characters that differ in two 8-character words
__device__ __forceinline__ int ndf_diff8(u64 a, u64 b) {
    u64 t = a ^ b;
    t |= t >> 4; t |= t >> 2; t |= t >> 1;
    return __popcll(t & 0x0101010101010101ull);
}

__device__ __forceinline__ bool ndf_near(const u64 *__restrict__ a, const u64 *__restrict__ b, int W, int d,
                                         const i32 *__restrict__ pos, int k) {
    int mm = 0;
    for (int j = 0; j < W; ++j) {
        mm += ndf_diff8(a[j], b[j]);
        if (mm > d) return false;
    }
    const u8 *ab = (const u8 *)a, *bb = (const u8 *)b;
    for (int j = 0; j < k; ++j)
        if (ab[pos[j]] != bb[pos[j]]) return false;  // different bucket (hash collision)
    return true;
}

__global__ void __launch_bounds__(256)
ndf_edge_kernel(const u64 *__restrict__ padded, u32 n, int W, int d, const i32 *__restrict__ pos, int k,
                const u64 *__restrict__ keys, const u32 *__restrict__ vals, u32 *__restrict__ e_i,
                u32 *__restrict__ e_j, u32 *__restrict__ count, u32 cap, const u32 *__restrict__ grp,
                size_t pos_group_stride, int n_earlier) {
    u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    u32 npairs = 0;
    if (x < n) {
        const u64 key = keys[x];
        const u32 i = vals[x];
        const u64 *a = padded + (size_t)i * W;
        if (grp) pos += (size_t)grp[i] * pos_group_stride;
        for (u32 y = x; y-- > 0;) {
            if (keys[y] != key) break;
            const u32 j = vals[y];  // j < i: stable sort keeps indices ascending in a run
            if (grp && grp[j] != grp[i]) continue;   // another group under the same key
            // (pos points at THIS table's positions; the earlier tables' lie k entries apart before it)
            const u8 *ab = (const u8 *)a, *bb = (const u8 *)(padded + (size_t)j * W);
            bool seen = false;
            for (int tp = 1; tp <= n_earlier && !seen; ++tp) {
                const i32 *pe = pos - (size_t)tp * k;
                bool eq = true;
                for (int q = 0; q < k && eq; ++q) eq = ab[pe[q]] == bb[pe[q]];
                seen = eq;      // the pair met in that table's bucket: compared there
            }
            if (seen) continue;
            ++npairs;
            if (ndf_near(a, padded + (size_t)j * W, W, d, pos, k)) {
                const u32 shard = (x >> 6) & (ES_SHARDS - 1);
                const u32 slot = atomicAdd(&count[shard * ES_STRIDE], 1u);
                if (slot < cap) { e_i[(size_t)shard * cap + slot] = i; e_j[(size_t)shard * cap + slot] = j; }
            }
        }
    }
    // pairs compared (SURVEY 8(d) K3's C): one 64-bit add per wavefront, beside its shard's edge counter
    for (int o = 32; o > 0; o >>= 1) npairs += __shfl_down(npairs, o, WAVE);
    if ((threadIdx.x & 63) == 0 && npairs)
        atomicAdd((unsigned long long *)(count + ((x >> 6) & (ES_SHARDS - 1)) * ES_STRIDE + 2), (unsigned long long)npairs);
}

// status: 0 undecided, 1 kept, 2 dropped.  flags: bit0 = has kept higher
// neighbour, bit1 = has undecided higher neighbour
__global__ void __launch_bounds__(256)
ndf_edge_round_kernel(const u32 *__restrict__ e_i, const u32 *__restrict__ e_j, u32 segcap,
                      const u32 *__restrict__ count, const u32 *__restrict__ status, u32 *__restrict__ flags) {
    const u64 t64 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 shard = (u32)(t64 / segcap);
    if (shard >= ES_SHARDS) return;
    if ((u32)(t64 - (u64)shard * segcap) >= min(count[shard * ES_STRIDE], segcap)) return;
    const size_t t = (size_t)t64;
    u32 i = e_i[t];
    if (status[i] != 0) return;
    u32 sj = status[e_j[t]];
    if (sj == 1) atomicOr(&flags[i], 1u);
    else if (sj == 0) atomicOr(&flags[i], 2u);
}

__global__ void __launch_bounds__(256)
ndf_node_round_kernel(u32 *__restrict__ status, u32 *__restrict__ flags, u32 n, u32 *__restrict__ undecided) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (status[i] != 0) return;
    u32 f = flags[i];
    flags[i] = 0;
    if (f & 1u) status[i] = 2;
    else if (!(f & 2u)) status[i] = 1;
    else atomicAdd(undecided, 1u);
}

// Lazy resolution (round 3).  Appending EVERY near pair of a run and resolving afterwards costs the square of
// the run length, and with hundreds or thousands of near-identical strains per species (S5) the runs are that
// long: 89 of the 106 s of the filters at S5 x 1.0.  But a probe's fate only depends on higher-priority mates
// that are KEPT (one of them near: the probe is dropped) or still UNDECIDED (one of them near: it has to wait);
// dropped mates never matter.  So every (table, sorted slot) keeps a cursor into its run, from the run's first
// (highest-priority) slot upwards, and a round moves it on: past dropped mates without looking at them, past
// mates that are not near (compared once, never again), and it stops at the first near mate that is not
// dropped -- kept: the probe is dropped on the spot (final, whatever the other tables say); undecided: the
// probe waits, the cursor remembers that this mate is near.  A probe none of whose cursors had to wait in a
// round, all of them at the end of their runs' prefixes, is kept (ndf_node_round_kernel).  Decisions only ever
// rest on decided, hence final, states, so the fixed point is the reference's sequential pass.  In a run of
// near-identical probes every slot compares with the run's head once and is dropped in the next round.
// A pair that shares a bucket in several tables belongs to the first of them (Family::owned_earlier): a probe
// is only kept once ALL its cursors have run out, so the pair is looked at there.
// An entry = table * n + slot; round 0 walks all of them (list == nullptr), table by table -- the blocks of a
// launch start roughly in order, so a probe that waits because of an early table is mostly not looked at
// again by the later ones -- and every round lists the entries that still have work to do.
// Long walks are shared: a slot deep in a run of thousands has thousands of dropped mates to pass, one
// dependent load each -- tens of milliseconds for ONE lane while the launch waits
// or, in a bucket of diverse probes, thousands of kept mates to compare with, microseconds each: tens of
// milliseconds for ONE lane while the launch waits (measured: rounds of 3,000 entries took as long as rounds
// of 900,000).  After a few steps of its own a lane therefore hands its walk to the wavefront: 64 mates per
// step, every lane looks at one (state, bucket, distance), the first near one decides.
#define NDF_CUR_NEAR 0x80000000u
#define NDF_CUR_NONE 0xffffffffu
#define NDF_OWN_STEPS 8
#define NDF_WAVE_NEAR_MAX 24
#define NDF_DRAIN_BLOCKS 1280u     // the drain launch: 256 CUs x 5 workgroups of 4 wavefronts (32 KB of LDS each)

struct HammingFamily {       // ndf_near on the padded rows; the earlier tables' sampled positions lie k entries apart
    static constexpr bool WAVE_NEAR = false;     // (a comparison is a few XORs and popcounts: nothing to share)
    static constexpr int wave_near_max = 0;
    static constexpr bool WAVE64 = false;
    struct Scratch { u32 unused; };
    __device__ __forceinline__ bool near_wave(u32, u32, Scratch &, u32) const { return false; }
    __device__ __forceinline__ bool wave64_stage(u32, Scratch &, u32) const { return false; }
    __device__ __forceinline__ int wave64_first_near(u32, unsigned long long, u32, Scratch &, u32, u32 &) const { return -1; }
    const u64 *padded;
    int W, d, k;
    const i32 *pos_all;
    const u32 *grp;
    size_t pos_group_stride;
    int dedupe;              // from four tables on (with two or three the look at the earlier positions costs more than it saves)
    __device__ __forceinline__ const i32 *pos(u32 t, u32 i) const { return pos_all + (size_t)t * k + (grp ? (size_t)grp[i] * pos_group_stride : 0); }
    __device__ __forceinline__ bool same_bucket(u32, u32 i, u32 j) const { return !grp || grp[i] == grp[j]; }   // (ndf_near checks the sampled characters)
    __device__ __forceinline__ bool owned_earlier(u32 t, u32 i, u32 j) const {
        if (!dedupe) return false;
        const u8 *ab = (const u8 *)(padded + (size_t)i * W), *bb = (const u8 *)(padded + (size_t)j * W);
        for (u32 tp = 0; tp < t; ++tp) {
            const i32 *pe = pos(tp, i);
            bool eq = true;
            for (int q = 0; q < k && eq; ++q) eq = ab[pe[q]] == bb[pe[q]];
            if (eq) return true;
        }
        return false;
    }
    __device__ __forceinline__ bool near(u32 t, u32 i, u32 j) const {
        return ndf_near(padded + (size_t)i * W, padded + (size_t)j * W, W, d, pos(t, i), k);
    }
};

// WAKE (round 4, the default): driven by wake-ups instead of polling.  The polling form lists ALL unexhausted entries
// of every undecided probe every round -- a probe that waits for an undecided mate has its 25 cursors looked at
// ~45 times (S5 x 1.0: 1.35e9 entry visits per chunk for 6.9e7 comparisons).  Here a probe that finds an undecided
// near mate j PARKS: wait_on[i] = j (flags is that array), none of its entries is listed again; after every pass one
// thread per probe looks at its blocker (ndf_wake_kernel): kept -> the probe is dropped; dropped -> the probe wakes
// up and is listed for the next pass, whose threads are (table, listed probe) pairs in TABLE-major order (the entry
// is found through inv[table][probe] = slot; table-major, because the blocks of a launch start roughly in order: a
// probe that parks because of an early table is mostly not walked by the later ones -- entry by entry, a woken
// probe's 25 cursors sat in one wavefront and all walked: 1.7 x the comparisons); undecided -> it keeps waiting.
// A probe is kept by the entry that exhausts its last table (left[i] counts them down) -- inside the pass, so
// chains resolve as far as the timing allows.  Every decision still rests on decided, hence final, states; waits
// point from a probe to a higher-priority one, so there are no cycles and the highest-priority undecided probe
// never waits: the same fixed point.  (Built once before the comparisons were shared by the wavefront: a fifth of
// the entry visits and the same time, because the rounds were ALU-bound then.)
// Deferred walks (round 5).  A wavefront of 64 entries used to finish its long walks one after the other: with a few
// hundred probes listed, 64 walks of tens of microseconds each sat in ONE wavefront while the device idled (a pass over
// 666 probes: 3 ms), and in the large passes the slowest wavefront set the time.  Now a pass has two launches: the
// first gives every entry its NDF_OWN_STEPS own steps and files the walks that are not over in a queue (dq; the cursor
// is saved); the second (drain) is a fixed grid of wavefronts that take the queued walks one by one through a ticket
// share -- one walk per wavefront at a time, 64 mates per step.  Any order of the walks gives the same fixed point
// (decisions rest on final states only).  The queue has ES_SHARDS shards (a counter on ONE address takes ~10 ns per
// atomic: 100 k walks filed or handed out through one counter were 1-2 ms of every pass); the wavefronts of the first
// launch file round-robin, shard s is drained by the wavefronts with (id & 63) == s, strided.
struct NdfQueue {
    u32 *dq; u32 *dq_count; u32 segcap;      // shard s: dq[s * segcap ..), its length dq_count[s * ES_STRIDE]
    u32 *overflow;                           // set (and the entry not filed) if a shard's segment were ever exceeded: the host then fails the call
    u32 *st2;                                // the states once more, 2 bits per probe (see ndf_state); null: polling rounds
};
// A mate's state is looked up once per mate examined, at random: the 4-byte words of 5 M probes are 18 MB that miss the
// L2 (PMC, round 5: ndf_cflag_kernel alone fetched 125 GB per S5 step for "is the probe at this slot dropped?").  The same
// states packed 16 to a word -- 1.1 MB, resident in every XCD's L2 -- are kept beside them: 0 undecided, 1 kept, 2 dropped
// only ever go 0 -> 1 or 0 -> 2, so a decision is an atomicOr; a reader that sees the old 0 sees what it would have seen a
// moment earlier.
__device__ __forceinline__ u32 ndf_state(const volatile u32 *st, const u32 *st2, u32 j) {
    return st2 ? (((const volatile u32 *)st2)[j >> 4] >> ((j & 15u) * 2u)) & 3u : st[j];
}
__device__ __forceinline__ void ndf_mark(u32 *st2, u32 i, u32 v) {
    if (st2) atomicOr(&st2[i >> 4], v << ((i & 15u) * 2u));
}

// Wavefronts per SIMD the register allocation of the two walking kernels aims at (round 6).  They wait on dependent looks
// (mate -> state -> signature -> codes), so what hides the waiting is how many wavefronts a SIMD holds: the MinHash instances
// took 118-126 and 82 vector registers = 4 and 5 wavefronts (the probe pass also held to 5 by 32 KB of LDS per workgroup).
// Per S5 step with one worker under rocprofv3, round-0 pass / probe passes: as compiled 486 / 424 ms; 5 / 6 wavefronts 354 /
// 325; 6 / 8: 273 / 284 (80 / 64 registers, ~100 bytes of spill); 8 / 8: 317 / 276.
#ifndef NDF_LAZY_WAVES
#define NDF_LAZY_WAVES 6
#endif
#ifndef NDF_PROBE_WAVES
#define NDF_PROBE_WAVES 8
#endif
template <class Family, bool WAKE, bool DRAIN = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NDF_LAZY_WAVES)))
ndf_lazy_kernel(Family fam, u32 n, const u64 *__restrict__ keys_all, const u32 *__restrict__ vals_all,
                u32 *__restrict__ cursor_all, u32 *status, u32 *flags, unsigned long long *__restrict__ pairs,
                const u32 *__restrict__ list, u32 nlist, u32 *__restrict__ next, u32 *__restrict__ next_count,
                u32 *__restrict__ left, const u32 *__restrict__ inv, u32 ntables, u32 nslot, u32 TS, bool wave64, NdfQueue Q) {
    // cursor_all, inv: probe-major, [probe][TS] (round 5: a probe's tables side by side)
    // nslot = slots per table: n at first, fewer once the dropped probes' slots have been compacted away (WAKE)
    const u32 lane = threadIdx.x & 63;
    const u32 wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const volatile u32 *st = status;
    __shared__ typename Family::Scratch s_scratch[4];
    typename Family::Scratch &scratch = s_scratch[threadIdx.x >> 6];
    constexpr bool drain = WAKE && DRAIN;
    for (u32 iter = 0;; ++iter) {
    bool again = false, walking = false, near_known = false, valid = false, deferred = false;
    u32 e = 0, t = 0, x = 0, i = 0, y = 0, compared = 0, found = 0;
    u32 verdict = 0;                                 // 1 run exhausted, 2 dropped, 3 waits at y
    if (drain) {
        const u32 shard = wave_id & (ES_SHARDS - 1u);
        const u32 w = (wave_id >> 6) + iter * ((gridDim.x * blockDim.x) >> 12);     // (grid wavefronts / ES_SHARDS per stride)
        if (w >= min(Q.dq_count[shard * ES_STRIDE], Q.segcap)) break;
        if (lane == 0) {
            valid = true;
            e = Q.dq[(size_t)shard * Q.segcap + w];
            t = e / nslot;
            x = e - t * nslot;
            i = vals_all[e];
        }
    } else {
        if (iter) break;
        const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
        if (g < ((WAKE && list) ? nlist * ntables : nlist)) {
            valid = true;
            if (WAKE && list) {
                t = g / nlist;
                i = list[g - t * nlist];
                x = inv[(size_t)i * TS + t];
                e = t * nslot + x;
            } else {
                e = list ? list[g] : g;
                t = e / nslot;
                x = e - t * nslot;
                // (round 0 of the wake-up form: the first slot of a run has nobody before it -- ndf_init_kernel wrote its
                // cursor as run out and left it out of the probe's count, so it costs two coalesced key loads here
                // instead of five scattered accesses)
                if (WAKE && !list && (x == 0u || keys_all[e - 1] != keys_all[e])) valid = false;
                else i = vals_all[(size_t)t * nslot + x];
            }
        }
    }
    if (valid) {
        const u64 *keys = keys_all + (size_t)t * nslot;
        const u32 cur = cursor_all[(size_t)i * TS + t];
        if (st[i] == 0 && cur != x) {
            again = true;
            // (polling: flags[i] != 0 = waiting already in this round; wake-ups: flags[i] = the blocker, parked)
            if (((const volatile u32 *)flags)[i] == (WAKE ? NDF_CUR_NONE : 0u)) {
                walking = true;
                if (cur == NDF_CUR_NONE) {           // first visit: the first slot of the run
                    const u64 key = keys[x];
                    u32 lo = 0, hi = x;
                    while (lo < hi) {
                        const u32 mid = (lo + hi) >> 1;
                        if (keys[mid] < key) lo = mid + 1; else hi = mid;
                    }
                    y = lo;
                } else {
                    near_known = (cur & NDF_CUR_NEAR) != 0;
                    y = cur & ~NDF_CUR_NEAR;
                }
            }
        }
    }
    const u32 *vals = vals_all + (size_t)t * nslot;
    // The comparisons the lanes of this wavefront want right now (lane: its probe pi against mate pj).  A comparison of
    // two k-mer sets is a merge walk of ~100 dependent steps, ~2,500 instructions -- and the whole wavefront executes
    // them while typically a handful of its lanes compare (PMC, round 4: 9,500 VALU instructions per wavefront;
    // the rounds are ALU-bound).  So with fewer than NDF_WAVE_NEAR_MAX lanes wanting one the wavefront runs them one
    // after the other TOGETHER (Family::near_wave: ~200 instructions each); with more, every lane runs its own.
    auto compare = [&](bool want, u32 tt, u32 pi, u32 pj) -> bool {
        bool is_near = false;
        unsigned long long wb = __ballot(want);
        if (!wb) return false;
        if (!Family::WAVE_NEAR || __popcll(wb) >= fam.wave_near_max) {
            if (want) is_near = fam.near(tt, pi, pj);
        } else {
            while (wb) {
                const int b = __ffsll((long long)wb) - 1;
                wb &= wb - 1ull;
                const bool r = fam.near_wave((u32)__shfl((int)pi, b, WAVE), (u32)__shfl((int)pj, b, WAVE), scratch, lane);
                if ((int)lane == b) is_near = r;
            }
        }
        return is_near;
    };
    // the mate at y (y < x) of the lanes in `act`: a verdict, or on to the next
    auto examine = [&](bool act) {
        u32 j = 0, sj = 2;
        bool is_near = false, want = false;
        if (act) {
            j = vals[y];                             // j < i: stable sort keeps indices ascending in a run
            sj = ndf_state(st, Q.st2, j);
            if (sj != 2) {
                is_near = near_known;
                want = !is_near && fam.same_bucket(t, i, j) && !fam.owned_earlier(t, i, j);
            }
        }
        const bool r = compare(want, t, i, j);
        if (want) { ++compared; is_near = r; found += r ? 1u : 0u; }
        if (act) {
            if (is_near) verdict = sj == 1 ? 2u : 3u;   // a kept higher-priority near-duplicate / an undecided one
            else { ++y; near_known = false; }
        }
    };
    if (!drain)
        for (int step = 0; step < NDF_OWN_STEPS; ++step) {
            bool act = walking && !verdict;
            if (act && y >= x) { verdict = 1; act = false; }
            if (!__ballot(act)) break;
            examine(act);
        }
    if (WAKE && Q.dq && !drain) {                    // the walks that are not over: into the queue
        if (walking && !verdict && y >= x) verdict = 1;
        deferred = walking && !verdict;
        const unsigned long long db = __ballot(deferred);
        if (db) {
            const u32 shard = wave_id & (ES_SHARDS - 1u);
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&Q.dq_count[shard * ES_STRIDE], (u32)__popcll(db));
            base = __shfl(base, 0, WAVE);
            if (deferred) {
                // (segcap = a shard's share + two wavefronts' worth: cannot be exceeded while the wavefronts file round
                // robin; guarded all the same -- ADVICE round 5 -- so that a change of the grid shape shows as an error
                // of the call instead of a write past the segment)
                const u32 at = base + (u32)__popcll(db & ((1ull << lane) - 1ull));
                if (at < Q.segcap) Q.dq[(size_t)shard * Q.segcap + at] = e; else *Q.overflow = 1u;
                cursor_all[(size_t)i * TS + t] = y | (near_known ? NDF_CUR_NEAR : 0u);
            }
        }
    }
    for (;;) {                                       // the walks that are not over yet, one at a time, by the whole wavefront
        const unsigned long long todo = __ballot(walking && !verdict && !deferred);
        if (!todo) break;
        const int leader = __ffsll((long long)todo) - 1;
        if (__shfl((int)(y >= x), leader, WAVE)) {    // (the solo steps ended right at the end of its run)
            if ((int)lane == leader) verdict = 1;
            continue;
        }
        if (__shfl((int)near_known, leader, WAVE)) {  // (its blocker first: one look)
            examine((int)lane == leader);
            continue;
        }
        const u32 ly = __shfl(y, leader, WAVE), lx = __shfl(x, leader, WAVE),
                  lt = (u32)__builtin_amdgcn_readfirstlane(__shfl((int)t, leader, WAVE)),
                  li = (u32)__builtin_amdgcn_readfirstlane(__shfl((int)i, leader, WAVE));
        const u32 *lvals = vals_all + (size_t)lt * nslot;
        u32 ny = ly, hit_state = 0, ncmp = 0, nfound = 0;
        // (MinHash: the walking probe's k-mers staged in LDS once, every comparison then by the whole wavefront)
        const bool staged = Family::WAVE64 && wave64 && fam.wave64_stage(li, scratch, lane);
        for (;;) {                                   // 64 mates per step, each lane looks at one
            const u32 yy = ny + lane;
            bool want = false;
            u32 sj = 2, j = 0;
            if (yy < lx) {
                j = lvals[yy];
                sj = ndf_state(st, Q.st2, j);
                want = sj != 2 && fam.same_bucket(lt, li, j) && !fam.owned_earlier(lt, li, j);
            }
            int first = -1;
            if (staged) {
                first = fam.wave64_first_near(li, __ballot(want), j, scratch, lane, ncmp);
                nfound += (first >= 0 && lane == 0) ? 1u : 0u;
            } else {
                const bool is_near = compare(want, lt, li, j);
                if (want) { ++ncmp; nfound += is_near ? 1u : 0u; }
                const unsigned long long m = __ballot(is_near);
                if (m) first = __ffsll((long long)m) - 1;
            }
            if (first >= 0) {
                ny += (u32)first;
                hit_state = __shfl(sj, first, WAVE);
                break;
            }
            ny += 64;
            if (ny >= lx) { ny = lx; break; }
        }
        compared += ncmp;                            // (every lane counts what it compared)
        found += nfound;
        if ((int)lane == leader) {
            y = ny;
            verdict = y >= x ? 1u : (hit_state == 1 ? 2u : 3u);
        }
    }
    if (walking && !deferred) {
        if (verdict == 2) { status[i] = 2; ndf_mark(Q.st2, i, 2u); again = false; }
        else if (verdict == 3) {                     // wait for the undecided mate at y
            if (WAKE) atomicCAS(&flags[i], NDF_CUR_NONE, vals[y]);      // (one blocker per probe: the first entry to find one)
            else flags[i] = 2;
        } else again = false;                        // this table has nothing more to say about i
        cursor_all[(size_t)i * TS + t] = verdict == 1 ? x : (y | NDF_CUR_NEAR);
        // the entry that exhausts a probe's last table: nobody kept is near -- kept
        if (WAKE && verdict == 1 && atomicSub(&left[i], 1u) == 1u && atomicCAS(&status[i], 0u, 1u) == 0u) ndf_mark(Q.st2, i, 1u);
    }
    if (__ballot(compared != 0u)) {                  // (one atomic per wavefront and counter)
        for (int o = 32; o > 0; o >>= 1) { compared += __shfl_xor(compared, o, WAVE); found += __shfl_xor(found, o, WAVE); }
        if (lane == 0) {
            atomicAdd(&pairs[wave_id & (ES_SHARDS - 1)], (unsigned long long)compared);
            if (found) atomicAdd(&pairs[ES_SHARDS + (wave_id & (ES_SHARDS - 1))], (unsigned long long)found);
        }
    }
    const unsigned long long bal = WAKE ? 0ull : __ballot(again);
    if (bal) {
        u32 base = 0;
        if (lane == 0) base = atomicAdd(next_count, (u32)__popcll(bal));
        base = __shfl(base, 0, WAVE);
        if (again) next[base + (u32)__popcll(bal & ((1ull << lane) - 1ull))] = e;
    }
    }
}

// One wavefront per listed probe, a lane per table (round 5; <= 64 tables).  The entry-per-thread pass above spends its
// time waiting: a woken probe's ~24 unexhausted cursors are 24 threads in different wavefronts, each with its own
// scattered loads of the slot, the cursor and the probe's state, and the comparisons the lanes of a wavefront want are
// run one after the other, each a chain of dependent loads (PMC: 13 % VALU, 40 % TA, ~80 us per wavefront).  Here the
// probe's cursors and slots are one row each (probe-major arrays, a cache line per probe), its k-mer codes are staged in
// LDS ONCE for all its tables' comparisons (Family::wave64_stage; four mates in flight, no dependent loads), and the
// tables are looked at in table order: the first table with a near mate that is not dropped decides (kept: the probe is
// dropped; undecided: it parks), the later tables are not compared -- what the table-major order of the entry pass
// achieved by timing.  A table whose walk is not over after NDF_PROBE_STEPS mates goes to the queue of deferred walks
// (64 mates per step there).
#define NDF_PROBE_STEPS 6
template <class Family>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NDF_PROBE_WAVES)))
ndf_probe_kernel(Family fam, u32 n, const u64 *__restrict__ keys_all, const u32 *__restrict__ vals_all,
                 u32 *__restrict__ cursor_all, u32 *status, u32 *flags, unsigned long long *__restrict__ pairs,
                 const u32 *__restrict__ list, u32 nlist, u32 *__restrict__ left, const u32 *__restrict__ inv,
                 u32 ntables, u32 nslot, u32 TS, bool wave64, NdfQueue Q) {
    const u32 lane = threadIdx.x & 63;
    const u32 wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const volatile u32 *st = status;
    __shared__ typename Family::Scratch s_scratch[4];
    typename Family::Scratch &scratch = s_scratch[threadIdx.x >> 6];
    if (wave_id >= nlist) return;
    const u32 i = (u32)__builtin_amdgcn_readfirstlane((int)(list ? list[wave_id] : wave_id));
    if (st[i] != 0u || ((const volatile u32 *)flags)[i] != NDF_CUR_NONE) return;      // (decided or parked meanwhile)
    const u32 t = lane;
    const bool active = t < ntables;
    const u32 x = active ? inv[(size_t)i * TS + t] : 0u;
    const u32 cur = active ? cursor_all[(size_t)i * TS + t] : x;
    const u64 *keys = keys_all + (size_t)(active ? t : 0u) * nslot;
    const u32 *vals = vals_all + (size_t)(active ? t : 0u) * nslot;
    bool walking = active && cur != x, near_known = false;
    u32 y = 0, exhausted = 0;           // exhausted: this lane's table ran out in this pass
    if (walking) {
        if (cur == NDF_CUR_NONE) {                   // first visit: the first slot of the run
            if (x == 0u || keys[x - 1u] != keys[x]) y = x;       // (alone in its bucket, or the run's head: one look)
            else {
                const u64 key = keys[x];
                u32 lo = 0, hi = x - 1u;
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    if (keys[mid] < key) lo = mid + 1; else hi = mid;
                }
                y = lo;
            }
        } else {
            near_known = (cur & NDF_CUR_NEAR) != 0;
            y = cur & ~NDF_CUR_NEAR;
        }
    }
    if (!__ballot(walking)) return;
    const bool staged = Family::WAVE64 && wave64 && fam.wave64_stage(i, scratch, lane);
    u32 compared = 0, found = 0;
    u32 outcome = 0, blocker = 0;       // wave-uniform: 0 nothing yet, 2 dropped, 3 parks on `blocker`
    for (int step = 0; step < NDF_PROBE_STEPS; ++step) {
        bool act = walking && !exhausted;
        if (act && y >= x) { exhausted = 1; act = false; }
        if (!__ballot(act)) break;
        u32 j = 0, sj = 2;
        bool is_near = false, want = false;
        if (act) {
            j = vals[y];                             // j < i: stable sort keeps indices ascending in a run
            sj = ndf_state(st, Q.st2, j);
            if (sj != 2) {
                is_near = near_known;
                want = !is_near && fam.same_bucket(t, i, j) && !fam.owned_earlier(t, i, j);
            }
        }
        // the tables in order: nothing beyond the first table with a known blocker needs a comparison
        const unsigned long long kb = __ballot(is_near);
        const unsigned long long below = kb ? ((1ull << (__ffsll((long long)kb) - 1)) - 1ull) : ~0ull;
        unsigned long long wb = __ballot(want) & below;
        int hit = -1;
        if (wb) {
            if (staged) {
                u32 ncmp = 0;
                hit = fam.wave64_first_near(i, wb, j, scratch, lane, ncmp);
                compared += ncmp;
                found += (hit >= 0 && lane == 0) ? 1u : 0u;
            } else {
                const bool mine = (wb >> lane) & 1ull;
                const bool r = mine ? fam.near(t, i, j) : false;
                if (mine) { ++compared; found += r ? 1u : 0u; }
                const unsigned long long m = __ballot(r);
                if (m) hit = __ffsll((long long)m) - 1;
            }
        }
        const int known = kb ? __ffsll((long long)kb) - 1 : 64;
        const int decisive = hit >= 0 ? hit : known;          // (hit < known by construction)
        if (decisive < 64) {
            // the lanes before it looked at a mate that is not near (or dropped): on; the decisive lane remembers its mate
            if (act && (int)lane < decisive) { ++y; near_known = false; }
            if ((int)lane == decisive) near_known = true;
            const u32 dsj = __shfl(sj, decisive, WAVE);
            blocker = __shfl(j, decisive, WAVE);
            outcome = dsj == 1u ? 2u : 3u;
            break;
        }
        if (act) { ++y; near_known = false; }
    }
    if (walking && !exhausted && y >= x) exhausted = 1;      // (also when the probe parks: a cursor at its own slot MEANS ran out, and is counted)
    if (outcome == 2u) {
        if (lane == 0) { status[i] = 2; ndf_mark(Q.st2, i, 2u); }
    } else {
        // cursors back; the tables that ran out are counted; the probe parks, or its unfinished walks are queued
        const unsigned long long eb = __ballot(exhausted != 0u);
        if (walking) cursor_all[(size_t)i * TS + t] = exhausted ? x : (y | (near_known ? NDF_CUR_NEAR : 0u));
        if (outcome == 3u) {
            if (lane == 0) flags[i] = blocker;       // (this wavefront is the only one that walks probe i in this launch)
        } else if (Q.dq) {
            const bool deferred = walking && !exhausted;
            const unsigned long long db = __ballot(deferred);
            if (db) {
                const u32 shard = wave_id & (ES_SHARDS - 1u);
                u32 base = 0;
                if (lane == 0) base = atomicAdd(&Q.dq_count[shard * ES_STRIDE], (u32)__popcll(db));
                base = __shfl(base, 0, WAVE);
                if (deferred) {
                    const u32 at = base + (u32)__popcll(db & ((1ull << lane) - 1ull));
                    if (at < Q.segcap) Q.dq[(size_t)shard * Q.segcap + at] = t * nslot + x; else *Q.overflow = 1u;
                }
            }
        }
        if (eb && lane == 0) {
            const u32 c = (u32)__popcll(eb);
            if (atomicSub(&left[i], c) == c && atomicCAS(&status[i], 0u, 1u) == 0u) ndf_mark(Q.st2, i, 1u);      // the last tables: nobody kept is near -- kept
        }
    }
    if (__ballot(compared != 0u)) {
        for (int o = 32; o > 0; o >>= 1) { compared += __shfl_xor(compared, o, WAVE); found += __shfl_xor(found, o, WAVE); }
        if (lane == 0) {
            atomicAdd(&pairs[wave_id & (ES_SHARDS - 1)], (unsigned long long)compared);
            if (found) atomicAdd(&pairs[ES_SHARDS + (wave_id & (ES_SHARDS - 1))], (unsigned long long)found);
        }
    }
}

__global__ void __launch_bounds__(256)
ndf_fill_u32_kernel(u32 *__restrict__ p, u32 n, u32 v) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// inv[table][probe] = the probe's slot in that table's sorted order
__global__ void __launch_bounds__(256)
ndf_inv_kernel(const u32 *__restrict__ vals_all, u32 TS, u32 nslot, size_t tn, u32 *__restrict__ inv) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tn) return;
    const size_t t = e / nslot;
    inv[(size_t)vals_all[e] * TS + t] = (u32)(e - t * nslot);
}
// inv, and round 5's head start: an entry that is the first of its run (or alone in its bucket: most entries) has
// nobody to look at -- its cursor is written as run out here (cursor == its own slot), the others count towards the probe's
// tables still to run out (left), and a probe without any is kept by ndf_keep_unshared_kernel
__global__ void __launch_bounds__(256)
ndf_init_kernel(const u32 *__restrict__ vals_all, const u64 *__restrict__ keys_all, u32 TS, u32 nslot, size_t tn,
                u32 *__restrict__ inv, u32 *__restrict__ cursor, u32 *__restrict__ left) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tn) return;
    const size_t t = e / nslot;
    const u32 x = (u32)(e - t * nslot), i = vals_all[e];
    inv[(size_t)i * TS + t] = x;
    if (x == 0u || keys_all[e - 1] != keys_all[e]) cursor[(size_t)i * TS + t] = x;
    else atomicAdd(&left[i], 1u);
}
__global__ void __launch_bounds__(256)
ndf_keep_unshared_kernel(u32 *__restrict__ status, const u32 *__restrict__ left, u32 n, u32 *__restrict__ st2) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && left[i] == 0u) { status[i] = 1u; ndf_mark(st2, i, 1u); }
}
// Compaction of the tables (WAKE): the slots of dropped probes go.  Every table holds every probe once, so all
// tables keep the same number of slots, and a probe deep in a run of thousands of near-identical strains -- all
// dropped after the second pass except a few -- no longer walks past them (64 per step, two dependent loads each: a
// pass that listed 1,469 probes took 4 ms because one of them crossed ~10^5 dropped mates; every pass had such a
// probe).  flag[e] = the slot's probe is not dropped; pos = exclusive scan of flag over ALL tables' slots (= table *
// new slots per table + rank inside the table).  Cursors are renumbered: past the dropped mates a cursor pointed at
// or before; one that reaches its own slot that way has exhausted its table (counted here, as the pass would).
__global__ void __launch_bounds__(256)
ndf_cflag_kernel(const u32 *__restrict__ vals_all, const u32 *__restrict__ status, size_t tn, u32 *__restrict__ flag,
                 const u32 *__restrict__ st2) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < tn) flag[e] = ndf_state(status, st2, vals_all[e]) != 2u ? 1u : 0u;
    else if (e == tn) flag[e] = 0u;
}
__global__ void __launch_bounds__(256)
ndf_compact_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ vals, u32 *__restrict__ cursor, u32 TS,
                   const u32 *__restrict__ flag, const u32 *__restrict__ pos, u32 nslot, u32 nslot_new, size_t tn,
                   u64 *__restrict__ keys2, u32 *__restrict__ vals2, u32 *status, u32 *__restrict__ left, u32 *__restrict__ st2) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tn || !flag[e]) return;
    const u32 t = (u32)(e / nslot), x = (u32)(e - (size_t)t * nslot);
    const u32 e2 = pos[e], x2 = e2 - t * nslot_new;
    const u32 i = vals[e];
    keys2[e2] = keys[e];
    vals2[e2] = i;
    u32 cur = cursor[(size_t)i * TS + t];                    // (probe-major: renumbered in place)
    if (cur == x) cur = x2;                                  // exhausted before (and counted)
    else if (cur != NDF_CUR_NONE) {
        const u32 y = cur & ~NDF_CUR_NEAR;
        const size_t ey = (size_t)t * nslot + y;
        const u32 y2 = pos[ey] - t * nslot_new;              // the first surviving mate at or after y
        if (y2 >= x2) {
            cur = x2;                                        // only dropped mates were left: this table is done with i
            if (atomicSub(&left[i], 1u) == 1u && atomicCAS(&status[i], 0u, 1u) == 0u) ndf_mark(st2, i, 1u);
        } else cur = y2 | (((cur & NDF_CUR_NEAR) && flag[ey]) ? NDF_CUR_NEAR : 0u);
    }
    cursor[(size_t)i * TS + t] = cur;
}

// after a pass: every undecided probe looks at its blocker; counters[1] = probes listed for the next pass, counters[0] =
// undecided probes (counted when the rounds are traced, otherwise just a flag that somebody is).  One atomic per
// WORKGROUP of 1,024 for the list (round 5: one per wavefront plus one for the count were 140 k atomics on ONE address
// each, 1.4 ms of every pass)
__global__ void __launch_bounds__(1024)
ndf_wake_kernel(u32 *status, u32 *__restrict__ wait_on, u32 n, u32 *__restrict__ next, u32 *__restrict__ counters, int count_undecided,
                u32 *__restrict__ st2) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ u32 s_cnt[16], s_und[16], s_base;
    __shared__ u32 s_drop;                       // probes this workgroup dropped (counters[2]: progress without a pass)
    if (threadIdx.x == 0) s_drop = 0;
    __syncthreads();
    bool undecided = false, woken = false;
    if (i < n && status[i] == 0) {
        const u32 j = wait_on[i];
        // (a probe that is neither decided nor parked after a pass cannot exist: every listed entry walks to a
        // verdict; the read-back checks that nothing is left undecided)
        undecided = true;
        if (j != NDF_CUR_NONE) {
            const u32 sj = ndf_state(status, st2, j);
            if (sj == 1) { status[i] = 2; ndf_mark(st2, i, 2u); undecided = false; s_drop = 1u; }
            else if (sj == 2) { wait_on[i] = NDF_CUR_NONE; woken = true; }
        }
    }
    const unsigned long long wb = __ballot(woken), ub = __ballot(undecided);
    if (lane == 0) { s_cnt[wv] = (u32)__popcll(wb); s_und[wv] = (u32)__popcll(ub); }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 tot = 0, und = 0;
        for (u32 w = 0; w < (blockDim.x >> 6); ++w) { const u32 c = s_cnt[w]; s_cnt[w] = tot; tot += c; und += s_und[w]; }
        s_base = tot ? atomicAdd(&counters[1], tot) : 0u;
        if (und) { if (count_undecided) atomicAdd(&counters[0], und); else if (((volatile u32 *)counters)[0] == 0u) counters[0] = 1u; }    // (a flag is enough for the loop)
        if (s_drop && ((volatile u32 *)counters)[2] == 0u) counters[2] = 1u;
    }
    __syncthreads();
    if (woken) next[s_base + s_cnt[wv] + (u32)__popcll(wb & ((1ull << lane) - 1ull))] = i;
}

// Where the verdicts go (round 5): a host array of bytes (the C-ABI entry points), or 0/1 flags on the device for
// callers that go on there (the candidates object: no 20-MB read-back and re-upload per call).
struct NdfKeep { u8 *host; u32 *d_flags; };
__global__ void __launch_bounds__(256)
ndf_flags_kernel(const u32 *__restrict__ status, u32 n, u32 *__restrict__ flags, u32 *__restrict__ unresolved) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 st = status[i];
    flags[i] = st == 1u ? 1u : 0u;
    if (st == 0u) atomicAdd(unresolved, 1u);
}
static int ndf_keep_out(catchhip_ctx *ctx, u32 nn, const u32 *d_status, NdfKeep keep) {
    hipStream_t s = ctx->stream;
    if (keep.d_flags) {
        DevBuf<u32> bad;
        TRY(bad.alloc(1));
        HIP_TRY(hipMemsetAsync(bad.p, 0, sizeof(u32), s));
        hipLaunchKernelGGL(ndf_flags_kernel, dim3((unsigned)div_up(nn, 256)), dim3(256), 0, s, d_status, nn, keep.d_flags, bad.p);
        HIP_TRY(hipMemcpyAsync(ctx->h_pin, bad.p, sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (*(volatile u32 *)ctx->h_pin) { chip_set_error("ndf: unresolved probe"); return CATCHHIP_EINVAL; }
        return 0;
    }
    std::vector<u32> h_status(nn);
    HIP_TRY(hipMemcpyAsync(h_status.data(), d_status, sizeof(u32) * nn, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (u32 i = 0; i < nn; ++i) {
        if (h_status[i] == 0) { chip_set_error("ndf: unresolved probe"); return CATCHHIP_EINVAL; }
        keep.host[i] = h_status[i] == 1 ? 1 : 0;
    }
    return 0;
}

// what the lazy resolution keeps resident: keys, values, cursors (16 B), the slot of every probe in every table (4 B),
// and, while the tables are compacted, a second copy and the scan's flags (24 B) per (table, probe) entry.  It is taken only when that fits what the device has free right now (plus this library's idle
// cache, which an allocation returns to the driver first): otherwise the table-by-table edge-list variant, which
// needs a few buffers of n entries (a failed allocation used to fail the filter instead)
static bool ndf_lazy_fits(size_t tn) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return true;     // (no information: try, as before)
    int64_t st[4] = {0, 0, 0, 0};
    (void)catchhip_pool_stats(st);
    const size_t idle = st[2] > 0 ? (size_t)st[2] : 0;
    return (double)tn * 64.0 + (double)(64u << 20) <= 0.9 * ((double)fr + (double)idle);
}

// the rounds of the lazy resolution and the read-back (both families): launch(list or nullptr, nlist, next,
// next_count) queues one pass over the listed entries
// launch(wake, list or nullptr, nlist, next, next_count, left, inv): one pass (polling: over the listed entries;
// wake-ups: over the tables of the listed probes)
// probe-major stride of the cursor / slot arrays: a power of two up to 16 tables, then whole cache lines
static u32 ndf_table_stride(u32 ntables) {
    if (ntables > 16) return (ntables + 31u) & ~31u;
    u32 ts = 1;
    while (ts < ntables) ts <<= 1;
    return ts;
}

// launch(wake, list, nlist, next, next_count, left, inv, nslot, Q, mode): mode 0 = one pass of ndf_lazy_kernel (polling:
// over the listed entries; wake-ups: over the tables of the listed probes), 1 = its drain launch over the queue of
// deferred walks, 2 = ndf_probe_kernel over the listed probes (nullptr: all of them)
template <class Launch>
static int ndf_lazy_rounds(catchhip_ctx *ctx, u32 nn, size_t tn, DevBuf<u32> &count, DevBuf<u32> &status, DevBuf<u32> &flags,
                           DevBuf<u64> &pairs, PhaseTimer &tm, NdfKeep keep, Launch launch, DevBuf<u64> &skeys, DevBuf<u32> &svals_buf,
                           DevBuf<u32> &cursor, u32 ntables) {
    const u32 *svals = svals_buf.p;
    hipStream_t s = ctx->stream;
    const unsigned nb = (unsigned)div_up(nn, 256);
    u32 *undecided = count.p + ES_SHARDS * ES_STRIDE;     // [0] undecided probes, [1] entries / probes listed for the next round
    const bool wake = !chip_test_env("CATCHHIP_NDF_POLL_ROUNDS");
    const u32 TS = ndf_table_stride(ntables);
    DevBuf<u32> lists[2], inv, left_tables, dq;
    TRY(lists[0].alloc(wake ? (size_t)nn : tn));
    TRY(lists[1].alloc(wake ? (size_t)nn : tn));
    // test hooks: the walks finished inside their wavefronts / the entry-per-thread pass in every round
    const bool queued = wake && !chip_test_env("CATCHHIP_NDF_NO_QUEUE");
    const bool probe_pass = queued && ntables <= 64 && !chip_test_env("CATCHHIP_NDF_NO_PROBE_PASS");
    const bool probe_round0 = probe_pass && chip_test_env("CATCHHIP_NDF_PROBE_ROUND0") != nullptr;
    NdfQueue Q{};
    DevBuf<u32> st2;
    if (wake && !chip_test_env("CATCHHIP_NDF_NO_PACKED_STATES")) {
        TRY(st2.alloc(((size_t)nn + 15) / 16 + 1));
        HIP_TRY(hipMemsetAsync(st2.p, 0, sizeof(u32) * (((size_t)nn + 15) / 16 + 1), s));
    }
    const u32 dq_segcap = (u32)(tn / ES_SHARDS) + 64u * 2u;      // (wavefronts file round-robin: a shard gets at most its share + one wavefront's)
    DevBuf<u32> dq_count, dq_overflow;
    if (queued) { TRY(dq.alloc((size_t)dq_segcap * ES_SHARDS)); TRY(dq_count.alloc(ES_SHARDS * ES_STRIDE)); }
    TRY(dq_overflow.alloc(1));
    HIP_TRY(hipMemsetAsync(dq_overflow.p, 0, sizeof(u32), s));
    if (wake) {
        TRY(inv.alloc((size_t)nn * TS));
        TRY(left_tables.alloc(nn));
        HIP_TRY(hipMemsetAsync(left_tables.p, 0, sizeof(u32) * nn, s));
        hipLaunchKernelGGL(ndf_init_kernel, dim3((unsigned)div_up((i64)tn, 256)), dim3(256), 0, s, svals, (const u64 *)skeys.p, TS, nn, tn,
                           inv.p, cursor.p, left_tables.p);
        hipLaunchKernelGGL(ndf_keep_unshared_kernel, dim3(nb), dim3(256), 0, s, status.p, (const u32 *)left_tables.p, nn, st2.p);
        HIP_TRY(hipMemsetAsync(flags.p, 0xff, sizeof(u32) * nn, s));       // (flags = wait_on: nobody is parked)
        tm.launch(2);
    } else {
        TRY(inv.alloc(1));
    }
    u32 left = nn, nlist = probe_round0 ? nn : (u32)tn, nslot = nn;
    const bool trace = getenv("CATCHHIP_TIMING") && atoi(getenv("CATCHHIP_TIMING")) > 1;
    const bool compacting = wake && !chip_test_env("CATCHHIP_NDF_NO_COMPACTION");
    DevBuf<u64> skeys2;
    DevBuf<u32> svals2, cflag, cpos, ctmp;
    auto t_round = std::chrono::steady_clock::now();
    for (u32 round = 0; left && round <= nn + 1; ++round) {
        // the tables without the dropped probes' slots: before the passes 2, 4, 8, 16, ... if a quarter or more would go
        if (compacting && round >= 2 && (round & (round - 1)) == 0 && nslot > 4096) {
            const size_t tcur = (size_t)ntables * nslot;
            TRY(cflag.reserve(tcur + 1));
            TRY(cpos.reserve(tcur + 1));
            hipLaunchKernelGGL(ndf_cflag_kernel, dim3((unsigned)div_up((i64)tcur + 1, 256)), dim3(256), 0, s, (const u32 *)svals_buf.p,
                               (const u32 *)status.p, tcur, cflag.p, (const u32 *)st2.p);
            TRY(chip_exclusive_scan_u32(ctx, cflag.p, cpos.p, (i64)tcur + 1, ctmp));
            HIP_TRY(hipMemcpyAsync(ctx->h_pin, cpos.p + tcur, sizeof(u32), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            const u32 total = *(volatile u32 *)ctx->h_pin;
            if (total % ntables != 0) { chip_set_error("ndf: tables disagree on the surviving probes"); return CATCHHIP_EINVAL; }
            const u32 nslot2 = total / ntables;
            if ((u64)nslot2 * 4 <= (u64)nslot * 3) {
                TRY(skeys2.reserve(tcur));
                TRY(svals2.reserve(tcur));
                hipLaunchKernelGGL(ndf_compact_kernel, dim3((unsigned)div_up((i64)tcur, 256)), dim3(256), 0, s, (const u64 *)skeys.p,
                                   (const u32 *)svals_buf.p, cursor.p, TS, (const u32 *)cflag.p, (const u32 *)cpos.p, nslot, nslot2,
                                   tcur, skeys2.p, svals2.p, status.p, left_tables.p, st2.p);
                skeys.swap(skeys2); svals_buf.swap(svals2);
                nslot = nslot2;
                if (nslot)
                    hipLaunchKernelGGL(ndf_inv_kernel, dim3((unsigned)div_up((i64)ntables * nslot, 256)), dim3(256), 0, s,
                                       (const u32 *)svals_buf.p, TS, nslot, (size_t)ntables * nslot, inv.p);
                tm.launch(4);
            }
        }
        HIP_TRY(hipMemsetAsync(undecided, 0, 3 * sizeof(u32), s));       // ([2]: the wake-up launch dropped somebody)
        if (queued) HIP_TRY(hipMemsetAsync(dq_count.p, 0, sizeof(u32) * ES_SHARDS * ES_STRIDE, s));
        Q.dq = queued ? dq.p : (u32 *)nullptr; Q.dq_count = dq_count.p; Q.segcap = dq_segcap; Q.st2 = st2.p; Q.overflow = dq_overflow.p;
        if (nlist) {
            const u32 *cur_list = round ? (const u32 *)lists[round & 1].p : (const u32 *)nullptr;
            const int mode = (probe_pass && (round || probe_round0)) ? 2 : 0;
            launch(wake, cur_list, nlist, lists[(round & 1) ^ 1].p, undecided + 1, left_tables.p, (const u32 *)inv.p, nslot, Q, mode);
            if (queued) {
                launch(wake, cur_list, nlist, lists[(round & 1) ^ 1].p, undecided + 1, left_tables.p, (const u32 *)inv.p, nslot, Q, 1);
                tm.launch(1);
            }
        }
        if (wake) hipLaunchKernelGGL(ndf_wake_kernel, dim3((unsigned)div_up(nn, 1024)), dim3(1024), 0, s, status.p, flags.p, nn, lists[(round & 1) ^ 1].p, undecided, trace ? 1 : 0, st2.p);
        else hipLaunchKernelGGL(ndf_node_round_kernel, dim3(nb), dim3(256), 0, s, status.p, flags.p, nn, undecided);
        tm.launch(2);
        HIP_TRY(hipMemcpyAsync(ctx->h_pin, undecided, 3 * sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (trace) {
            const auto t1 = std::chrono::steady_clock::now();
            u64 hp[2 * ES_SHARDS], cmp = 0;
            (void)hipMemcpy(hp, pairs.p, sizeof(hp), hipMemcpyDeviceToHost);
            for (int sh = 0; sh < ES_SHARDS; ++sh) cmp += hp[sh];
            fprintf(stderr, "[catchhip]     lazy round %u: %u entries -> %u undecided probes, %u entries next; %.2f ms, %llu pairs compared so far\n",
                    round, nlist, ((volatile u32 *)ctx->h_pin)[0], ((volatile u32 *)ctx->h_pin)[1],
                    std::chrono::duration<double, std::milli>(t1 - t_round).count(), (unsigned long long)cmp);
            t_round = std::chrono::steady_clock::now();
        }
        const bool idle = wake && !nlist;             // (no pass in this round)
        left = ((volatile u32 *)ctx->h_pin)[0];       // (wake-ups: a flag unless the rounds are traced)
        nlist = ((volatile u32 *)ctx->h_pin)[1];
        // A round without a pass in which the wake-up launch neither woke nor dropped anybody: whoever is undecided now
        // neither waits nor walks (a broken invariant).  (A round that only DROPS is progress: a chain of probes each
        // parked on the next resolves one link per wake-up launch -- the first form of this check stopped there: fuzz
        // seed 31188, 4,720 of 5,608 probes of near-identical small groups undecided after round 0.)
        if (idle && left && !nlist && !((volatile u32 *)ctx->h_pin)[2]) {
            chip_set_error("ndf: undecided probes that nobody will wake");
            return CATCHHIP_EINVAL;
        }
    }
    HIP_TRY(hipGetLastError());
    if (left) { chip_set_error("ndf: %u probes undecided after %u rounds", left, nn + 2); return CATCHHIP_EINVAL; }
    tm.stop();
    std::vector<u64> h_pairs(2 * ES_SHARDS);
    u32 h_overflow = 0;
    HIP_TRY(hipMemcpyAsync(h_pairs.data(), pairs.p, sizeof(u64) * 2 * ES_SHARDS, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&h_overflow, dq_overflow.p, sizeof(u32), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    tm.finish();
    if (h_overflow) { chip_set_error("ndf: the queue of deferred walks overflowed a shard's segment"); return CATCHHIP_EINVAL; }
    // pairs compared / of them near (the all-pairs variant reports the length of its edge list there)
    for (int sh = 0; sh < ES_SHARDS; ++sh) { ctx->ndf_counters[2] += (i64)h_pairs[sh]; ctx->ndf_counters[3] += (i64)h_pairs[ES_SHARDS + sh]; }
    return ndf_keep_out(ctx, nn, status.p, keep);
}

// greedy resolution rounds over the edge list + read-back (shared by the two
// LSH families)
// edges in the fullest shard
static int ndf_fullest_shard(catchhip_ctx *ctx, const u32 *count, u32 *out) {
    TRY(chip_pinned_reserve(ctx, sizeof(u32) * ES_WORDS));
    HIP_TRY(hipMemcpyAsync(ctx->h_big, count, sizeof(u32) * ES_WORDS, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    u32 m = 0;
    i64 edges = 0, pairs = 0;
    for (int sh = 0; sh < ES_SHARDS; ++sh) {
        const u32 v = ((const volatile u32 *)ctx->h_big)[sh * ES_STRIDE];
        if (v > m) m = v;
        edges += v;
        pairs += (i64)((const volatile unsigned long long *)((const u32 *)ctx->h_big + sh * ES_STRIDE + 2))[0];
    }
    ctx->ndf_counters[2] = pairs;   // pairs sharing a bucket that were compared (all tables)
    ctx->ndf_counters[3] = edges;   // of them within the distance
    *out = m;
    return 0;
}

// segused: edges in the fullest shard (the round kernel only looks at that many slots per shard)
static int ndf_resolve(catchhip_ctx *ctx, u32 nn, u32 segused, u32 segcap, DevBuf<u32> &e_i, DevBuf<u32> &e_j,
                       DevBuf<u32> &count, DevBuf<u32> &status, DevBuf<u32> &flags, PhaseTimer &tm, NdfKeep keep) {
    hipStream_t s = ctx->stream;
    const unsigned nb = (unsigned)div_up(nn, 256);
    u32 *undecided = count.p + ES_SHARDS * ES_STRIDE;
    (void)segused;
    for (u32 round = 0; round <= nn + 1; ++round) {
        HIP_TRY(hipMemsetAsync(undecided, 0, sizeof(u32), s));
        if (segused)
            hipLaunchKernelGGL(ndf_edge_round_kernel, dim3((unsigned)div_up((i64)ES_SHARDS * segcap, 256)), dim3(256), 0,
                               s, e_i.p, e_j.p, segcap, (const u32 *)count.p, status.p, flags.p);
        hipLaunchKernelGGL(ndf_node_round_kernel, dim3(nb), dim3(256), 0, s, status.p, flags.p, nn, undecided);
        tm.launch(2);
        HIP_TRY(hipMemcpyAsync(ctx->h_pin, undecided, sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (*(volatile u32 *)ctx->h_pin == 0) break;
    }
    tm.stop();
    HIP_TRY(hipStreamSynchronize(s));
    tm.finish();
    return ndf_keep_out(ctx, nn, status.p, keep);
}

// the filter on probes whose characters are already on the device (n rows of L)
// d_grp / ngroups: optional group of every row (device array); positions is then
// [ngroups][ntables][k], every group with its own sampled positions
int chip_ndf_hamming_device(catchhip_ctx *ctx, const u8 *d_rows, i64 n, i32 L, const i32 *positions, i32 ntables,
                            i32 k, i32 dist_thres, u8 *keep_host, const u32 *d_grp, i64 ngroups, u32 *d_keep_flags) {
    ARG_CHECK(ctx && n >= 0 && L > 0 && ntables >= 1 && k >= 1 && positions);
    if (n == 0) return 0;
    ARG_CHECK(d_rows && (keep_host || d_keep_flags));
    const NdfKeep keep{keep_host, d_keep_flags};
    ARG_CHECK(n < ((i64)1 << 31) && n * (i64)L < ((i64)1 << 40));
    if (!d_grp) ngroups = 1;
    ARG_CHECK(ngroups >= 1);
    for (i64 t = 0; t < ngroups * ntables * k; ++t) ARG_CHECK(positions[t] >= 0 && positions[t] < L);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const u32 nn = (u32)n;
    struct { const u8 *p; } d_bytes = {d_rows};
    DevBuf<i32> d_pos;
    DevBuf<u64> keys, keys_alt;
    DevBuf<u32> vals, vals_alt, e_i, e_j, count, status, flags;
    TRY(d_pos.alloc((size_t)ngroups * ntables * k));
    TRY(keys.alloc(nn));
    TRY(vals.alloc(nn));
    TRY(count.alloc(ES_WORDS + 8));
    TRY(status.alloc(nn));
    TRY(flags.alloc(nn));
    HIP_TRY(hipMemcpyAsync(d_pos.p, positions, sizeof(i32) * (size_t)ngroups * ntables * k, hipMemcpyHostToDevice, s));
    const size_t pstride = (size_t)ntables * k;
    HIP_TRY(hipMemsetAsync(count.p, 0, ES_WORDS * sizeof(u32), s));
    HIP_TRY(hipMemsetAsync(status.p, 0, sizeof(u32) * nn, s));
    HIP_TRY(hipMemsetAsync(flags.p, 0, sizeof(u32) * nn, s));
    const int W = (L + 7) / 8;
    DevBuf<u64> padded;
    TRY(padded.alloc((size_t)n * W));
    hipLaunchKernelGGL(ndf_pad_kernel, dim3((unsigned)div_up((i64)n * W * 8, 256)), dim3(256), 0, s,
                       (const u8 *)d_bytes.p, (u32)n, (int)L, W * 8, (u8 *)padded.p);

    PhaseTimer tm(ctx, PHASE_NDF);
    ctx->ndf_counters[0] = n; ctx->ndf_counters[1] = ntables; ctx->ndf_counters[2] = ctx->ndf_counters[3] = 0;
    const unsigned nb = (unsigned)div_up(nn, 256);
    if ((i64)ntables * n <= ((i64)1 << 31) && !chip_test_env("CATCHHIP_NDF_ALL_PAIRS") && ndf_lazy_fits((size_t)ntables * nn)) {
        // lazy resolution (ndf_lazy_kernel): all tables' sorted runs stay resident, one cursor per (table, slot) --
        // 24 bytes per entry, at most 2^31 entries (48 GB); beyond that the edge-list variant below, table by table
        DevBuf<u64> skeys, pairs;
        DevBuf<u32> svals, cursor;
        const size_t tn = (size_t)ntables * nn;
        TRY(skeys.alloc(tn));
        TRY(svals.alloc(tn));
        const u32 TS = ndf_table_stride((u32)ntables);
        TRY(cursor.alloc((size_t)nn * TS));
        TRY(pairs.alloc(2 * ES_SHARDS));
        HIP_TRY(hipMemsetAsync(cursor.p, 0xff, sizeof(u32) * (size_t)nn * TS, s));
        HIP_TRY(hipMemsetAsync(pairs.p, 0, sizeof(u64) * 2 * ES_SHARDS, s));
        for (int t = 0; t < ntables; ++t) {
            hipLaunchKernelGGL(ndf_key_kernel, dim3(nb), dim3(256), 0, s, d_bytes.p, nn, (int)L,
                               d_pos.p + (size_t)t * k, (int)k, keys.p, vals.p, d_grp, pstride, 32);
            TRY(chip_radix_sort_pairs(ctx, keys, keys_alt, vals, vals_alt, nn, 32));
            HIP_TRY(hipMemcpyAsync(skeys.p + (size_t)t * nn, keys.p, sizeof(u64) * nn, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(svals.p + (size_t)t * nn, vals.p, sizeof(u32) * nn, hipMemcpyDeviceToDevice, s));
            tm.launch(1 + 24 + 2);
        }
        HammingFamily fam{(const u64 *)padded.p, W, (int)dist_thres, (int)k, (const i32 *)d_pos.p, d_grp, pstride, ntables >= 4 ? 1 : 0};
        return ndf_lazy_rounds(ctx, nn, tn, count, status, flags, pairs, tm, keep,
                               [&](bool wake, const u32 *list, u32 nlist, u32 *next, u32 *next_count, u32 *left_tables, const u32 *inv, u32 nslot, NdfQueue Q, int mode) {
            if (wake && mode == 2)
                hipLaunchKernelGGL((ndf_probe_kernel<HammingFamily>), dim3((unsigned)div_up((i64)nlist * 64, 256)), dim3(256), 0, s,
                                   fam, nn, (const u64 *)skeys.p, (const u32 *)svals.p, cursor.p, status.p, flags.p,
                                   (unsigned long long *)pairs.p, list, nlist, left_tables, inv, (u32)ntables, nslot, TS, false, Q);
            else if (wake && mode == 1)
                hipLaunchKernelGGL((ndf_lazy_kernel<HammingFamily, true, true>), dim3(mode == 1 ? NDF_DRAIN_BLOCKS : (unsigned)div_up(list ? (i64)nlist * ntables : (i64)nlist, 256)), dim3(256), 0, s,
                                   fam, nn, (const u64 *)skeys.p, (const u32 *)svals.p, cursor.p, status.p, flags.p,
                                   (unsigned long long *)pairs.p, list, nlist, next, next_count, left_tables, inv, (u32)ntables, nslot, TS, false, Q);
                        else if (wake)
                hipLaunchKernelGGL((ndf_lazy_kernel<HammingFamily, true>), dim3(mode == 1 ? NDF_DRAIN_BLOCKS : (unsigned)div_up(list ? (i64)nlist * ntables : (i64)nlist, 256)), dim3(256), 0, s,
                                   fam, nn, (const u64 *)skeys.p, (const u32 *)svals.p, cursor.p, status.p, flags.p,
                                   (unsigned long long *)pairs.p, list, nlist, next, next_count, left_tables, inv, (u32)ntables, nslot, TS, false, Q);
            else
                hipLaunchKernelGGL((ndf_lazy_kernel<HammingFamily, false>), dim3((unsigned)div_up((i64)nlist, 256)), dim3(256), 0, s, fam, nn,
                                   (const u64 *)skeys.p, (const u32 *)svals.p, cursor.p, status.p, flags.p,
                                   (unsigned long long *)pairs.p, list, nlist, next, next_count, left_tables, inv, (u32)ntables, nslot, TS, false, Q);
        }, skeys, svals, cursor, (u32)ntables);
    }
    u32 cap = (u32)std::max<i64>((i64)1 << 14, std::min<i64>(n * 16, (i64)1 << 28) / ES_SHARDS);   // per shard
    u32 ne = 0;
    for (int attempt = 0;; ++attempt) {
        TRY(e_i.reserve((size_t)cap * ES_SHARDS));
        TRY(e_j.reserve((size_t)cap * ES_SHARDS));
        HIP_TRY(hipMemsetAsync(count.p, 0, ES_WORDS * sizeof(u32), s));
        for (int t = 0; t < ntables; ++t) {
            hipLaunchKernelGGL(ndf_key_kernel, dim3(nb), dim3(256), 0, s, d_bytes.p, nn, (int)L,
                               d_pos.p + (size_t)t * k, (int)k, keys.p, vals.p, d_grp, pstride);
            TRY(chip_radix_sort_pairs(ctx, keys, keys_alt, vals, vals_alt, nn, 64));
            hipLaunchKernelGGL(ndf_edge_kernel, dim3(nb), dim3(256), 0, s, (const u64 *)padded.p, nn, W,
                               (int)dist_thres, d_pos.p + (size_t)t * k, (int)k, keys.p, vals.p, e_i.p,
                               e_j.p, count.p, cap, d_grp, pstride,
                               // (with two or three tables the look at the earlier tables' positions costs more than
                               // the comparisons it saves: S3, 2 tables, 15.1 -> 18.0 ms)
                               (ntables < 4 || chip_test_env("CATCHHIP_MH_NO_DEDUPE")) ? 0 : t);
            tm.launch(2 + 24);
        }
        HIP_TRY(hipGetLastError());
        TRY(ndf_fullest_shard(ctx, count.p, &ne));
        if (ne <= cap) break;
        if (attempt >= 2) { chip_set_error("ndf: edge buffer overflow"); return CATCHHIP_ENOMEM; }
        cap = ne;
    }
    return ndf_resolve(ctx, nn, ne, cap, e_i, e_j, count, status, flags, tm, keep);
}

extern "C" int catchhip_ndf_hamming(catchhip_ctx *ctx, const u8 *bytes, i64 n, i32 L, const i32 *positions,
                                    i32 ntables, i32 k, i32 dist_thres, u8 *keep) {
    ARG_CHECK(ctx && n >= 0 && L > 0);
    PoolScope pool_scope(ctx);
    if (n == 0) return 0;
    ARG_CHECK(bytes && keep && n * (i64)L < ((i64)1 << 40));
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<u8> d_bytes;
    TRY(d_bytes.alloc((size_t)n * L));
    HIP_TRY(hipMemcpyAsync(d_bytes.p, bytes, (size_t)n * L, hipMemcpyHostToDevice, ctx->stream));
    return chip_ndf_hamming_device(ctx, d_bytes.p, n, L, positions, ntables, k, dist_thres, keep, nullptr, 1, nullptr);
}

// ------------------------------------------------------------------------
// MinHash family (catch/filter/near_duplicate_filter.py:148-190,
// catch/utils/lsh.py:48-215 with N = 1 and use_fast_str_hash = True).
//
// A hash function of the family is  h(s) = min over the k-mers x of s of
// (a * |hash(x)| + b) mod (2^31 - 1)  where hash() is the interpreter's str
// hash: SipHash-2-4 of the characters in CPython <= 3.10, keyed by the
// process-wide secret -- all-zero under PYTHONHASHSEED=0, the only setting
// under which the reference's filter is reproducible; that is the hash
// computed here (oracle/catch_oracle.c restates it, tests pin both to the
// interpreter and to vectors recorded from the reference).  A table key is the
// tuple of k such minima; two probes are near-duplicates when they share a key
// in some table and the Jaccard distance of their k-mer SETS is <= dist_thres
// (float64, evaluated exactly as the reference's expression).  The sequential
// pass is the same maximal-independent-set resolution as for Hamming.
//   mh_kmer_kernel      one wavefront per probe: |hash| mod p of every k-mer
//                       and the sorted list of its DISTINCT k-mers (the bytes
//                       themselves, <= 16 per k-mer in two u64 words; rank
//                       sort in LDS)
//   mh_keys_all_kernel  one wavefront per probe, all tables (in chunks): the k
//                       minima of every table -> signature, 64-bit grouping key
//   (radix sort)        buckets = runs of equal keys, indices ascending
//   mh_edge_kernel      one lane per sorted slot walks left over its run:
//                       same signature (exact) and Jaccard distance <= d
// ------------------------------------------------------------------------
#define MH_P 2147483647ull
#define MH_MAXK 256   // k-mers per probe one wavefront sorts
// A 2,048-bit fingerprint of a probe's k-mer set (round 4): every distinct k-mer sets one bit; excess = distinct
// k-mers minus bits set.  |A and B| <= popcount(bits A & bits B) + min(excess A, excess B): a pair whose bound is
// below the intersection the threshold needs is not near -- 32 words instead of a merge walk of ~100 dependent steps.
// (With 512 bits the bound sat ~15 above the intersection and rejected next to nothing: mates of one bucket share
// 30-50 of their ~90 k-mers and the threshold needs 52.)
#define MH_FPW 32
#define MH_TAB_BITS 10
#define MH_TAB (1u << MH_TAB_BITS)   // slots of the staged probe's k-mer table (<= MH_MAXK = 256 codes)
__device__ __forceinline__ u32 mh_fp_bit(u64 h, u64 l) {
    u64 x = (l ^ (h * 0x9e3779b97f4a7c15ull)) * 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    return (u32)(x * 0xc4ceb9fe1a85ec53ull >> 53);      // 11 bits
}

#define MH_KC_NONE 0xffffffffu
__device__ __forceinline__ u32 mh_base2(u8 ch) { return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u; }

// CPython <= 3.10 str hash of `len` ASCII characters, zero key (see above; internal.h)
__device__ __forceinline__ long long mh_pyhash(const u8 *__restrict__ src, int len) { return chip_pyhash_seed0(src, len); }

// does any probe hold a character outside A/C/G/T?  (then the 16-byte k-mer ids are kept for every probe; otherwise the
// 2-bit codes are all there is: 16 B x 91 k-mers x 5 M probes of writes and as much memory less per call)
__global__ void __launch_bounds__(256)
mh_other_chars_kernel(const u8 *__restrict__ bytes, u64 total, u32 *__restrict__ flag) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    bool other = false;
    for (u64 at = (u64)blockIdx.x * blockDim.x + threadIdx.x; at < total; at += stride) other = other || mh_base2(bytes[at]) > 3u;
    if (__ballot(other) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// |x| mod (2^31 - 1) of a 64-bit hash by Mersenne folds (2^31 = 1 mod p) instead of the compiler's 64-bit division by a
// constant (~30 instructions)
__device__ __forceinline__ u32 mh_abs_mod_p(long long h) {
    const u64 x = (u64)(h < 0 ? -h : h);                                     // (never LLONG_MIN: |hash| < 2^63 ... or 2^63 itself, which folds the same)
    u64 v = (x & MH_P) + ((x >> 31) & MH_P) + (x >> 62);                      // < 2^32 + 2
    u32 t = (u32)(v & MH_P) + (u32)(v >> 31);                                 // < 2^31 + 2
    return t >= (u32)MH_P ? t - (u32)MH_P : t;
}

// KS: the k-mer size at compile time (0: `ks` at run time).  Round 6: with KS known the character loops and SipHash's
// block loop unroll (the run-time form spent ~12 instructions per character on loop control and shifts by a counter),
// |hash| mod p is two Mersenne folds, and the rank sort of a probe of A/C/G/T only compares ONE 32-bit word per pair --
// code << 8 | slot, unique, so that the tie-break of equal k-mers is in the word: a v_cmp + v_addc per (y, slot) with
// the y's read four at a time as an LDS broadcast (KS <= 12: 24 bits of code + 8 bits of slot).
template <int KS>
__global__ void __launch_bounds__(64)
mh_kmer_kernel(const u8 *__restrict__ bytes, const u32 *__restrict__ probe_off, const u32 *__restrict__ koff, u32 n,
               int ks_rt, u32 *__restrict__ xs, u64 *__restrict__ id_hi, u64 *__restrict__ id_lo,
               u32 *__restrict__ nuniq, unsigned long long *__restrict__ fp, u32 *__restrict__ fp_excess,
               u32 *__restrict__ kc, u32 kstride) {
    __shared__ u64 s_hi[MH_MAXK], s_lo[MH_MAXK];
    __shared__ __attribute__((aligned(16))) u32 s_code[MH_MAXK + 4];
    __shared__ unsigned long long s_fp[MH_FPW];
    const u32 lane = threadIdx.x;
    const int ks = KS ? KS : ks_rt;
    constexpr bool WORDKEY = KS > 0 && KS <= 12;      // (code << 8 | slot fits 32 bits; slots < MH_MAXK = 256)
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        const u8 *p = bytes + probe_off[i];
        const u32 k0 = koff[i], nk = koff[i + 1] - k0;
        // packed form (kc, round 5): a probe of A/C/G/T only has its distinct k-mers as 2-bit codes as well (<= 15
        // characters, so that 0xffffffff is free to mean "nothing"); any other character: the row starts with MH_KC_NONE
        bool bad = ks > 15;
        for (u32 j = lane; j < nk + (u32)ks - 1u; j += 64) bad = bad || mh_base2(p[j]) > 3u;
        const bool packable = !__ballot(bad);
        __syncthreads();
        for (u32 j = lane; j < nk; j += 64) {
            const long long h = mh_pyhash(p + j, ks);
            xs[k0 + j] = mh_abs_mod_p(h);
            u64 lo = 0, hi = 0;   // the k-mer's bytes, big endian: order = string order
            u32 code = 0;         // (A < C < G < T in the codes as in the bytes: the same order)
#pragma unroll
            for (int c = 0; c < ks; ++c) {
                const u64 ch = p[j + c];
                if (c < ks - 8) hi = (hi << 8) | ch; else lo = (lo << 8) | ch;
                code = (code << 2) | (mh_base2((u8)ch) & 3u);
            }
            if (ks <= 8) hi = 0;
            s_hi[j] = hi; s_lo[j] = lo; s_code[j] = (WORDKEY && packable) ? (code << 8) | j : code;
        }
        if (WORDKEY && packable)
            for (u32 j = nk + lane; j < ((nk + 3u) & ~3u); j += 64) s_code[j] = 0xffffffffu;    // (the last broadcast of four: never less than a key)
        __syncthreads();
        // rank sort; equal k-mers get consecutive ranks, the first of each group is kept.  By the 32-bit codes when the
        // probe has them (round 5: the 128-bit comparisons of all four register slots were 3,300 of the kernel's ~4,500
        // instructions per probe), and only over the slots the probe fills
        u64 mh[MH_MAXK / 64], ml[MH_MAXK / 64];
        u32 mc[MH_MAXK / 64], rk[MH_MAXK / 64];
#pragma unroll
        for (int q = 0; q < MH_MAXK / 64; ++q) {
            const u32 j = lane + q * 64;
            rk[q] = 0;
            mh[q] = j < nk ? s_hi[j] : 0; ml[q] = j < nk ? s_lo[j] : 0; mc[q] = j < nk ? s_code[j] : 0;
        }
        if (WORDKEY && packable) {
            // (the slot count decided outside the loop: a probe of 100 characters has 91 k-mers of 10 = two slots)
#define MH_RANK_LOOP(NQ)                                                                                              \
            for (u32 y = 0; y < nk; y += 4) {                                                                         \
                const uint4 c = *(const uint4 *)&s_code[y];                                                           \
                _Pragma("unroll") for (int q = 0; q < (NQ); ++q)                                                      \
                    rk[q] += (u32)(c.x < mc[q]) + (u32)(c.y < mc[q]) + (u32)(c.z < mc[q]) + (u32)(c.w < mc[q]);       \
            }
            if (nk <= 64) { MH_RANK_LOOP(1) }
            else if (nk <= 128) { MH_RANK_LOOP(2) }
            else { MH_RANK_LOOP(MH_MAXK / 64) }
#undef MH_RANK_LOOP
#pragma unroll
            for (int q = 0; q < MH_MAXK / 64; ++q) mc[q] >>= 8;          // the plain codes from here on
        } else if (packable) {
            for (u32 y = 0; y < nk; ++y) {
                const u32 c = s_code[y];
#pragma unroll
                for (int q = 0; q < MH_MAXK / 64; ++q) {
                    if ((u32)q * 64u >= nk) break;
                    const u32 j = lane + q * 64;
                    rk[q] += (c < mc[q] || (c == mc[q] && y < j)) ? 1u : 0u;
                }
            }
        } else {
            for (u32 y = 0; y < nk; ++y) {
                const u64 h = s_hi[y], l = s_lo[y];
#pragma unroll
                for (int q = 0; q < MH_MAXK / 64; ++q) {
                    if ((u32)q * 64u >= nk) break;
                    const u32 j = lane + q * 64;
                    const bool less = h < mh[q] || (h == mh[q] && (l < ml[q] || (l == ml[q] && y < j)));
                    rk[q] += less ? 1u : 0u;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MH_MAXK / 64; ++q) {
            const u32 j = lane + q * 64;
            if (j < nk) { s_hi[rk[q]] = mh[q]; s_lo[rk[q]] = ml[q]; s_code[rk[q]] = mc[q]; }
        }
        if (lane < MH_FPW) s_fp[lane] = 0ull;
        __syncthreads();
        u32 out = 0;
        for (u32 c0 = 0; c0 < nk; c0 += 64) {
            const u32 j = c0 + lane;
            const bool first = j < nk && (j == 0 || s_hi[j] != s_hi[j - 1] || s_lo[j] != s_lo[j - 1]);
            const unsigned long long bal = __ballot(first);
            if (first) {
                const u32 o = out + (u32)__popcll(bal & ((1ull << lane) - 1ull));
                if (id_hi) { id_hi[k0 + o] = s_hi[j]; id_lo[k0 + o] = s_lo[j]; }
                const u32 bit = mh_fp_bit(s_hi[j], s_lo[j]);
                atomicOr(&s_fp[bit >> 6], 1ull << (bit & 63u));
                if (kc) kc[(size_t)i * kstride + o] = packable ? s_code[j] : MH_KC_NONE;
            }
            out += (u32)__popcll(bal);
        }
        if (kc)
            for (u32 q = out + lane; q < kstride; q += 64) kc[(size_t)i * kstride + q] = MH_KC_NONE;
        __syncthreads();
        if (lane < MH_FPW) fp[(size_t)i * MH_FPW + lane] = s_fp[lane];
        u32 set = lane < MH_FPW ? (u32)__popcll(s_fp[lane]) : 0u;
        for (int d = 32; d > 0; d >>= 1) set += __shfl_xor(set, d, WAVE);
        if (lane == 0) { nuniq[i] = out; fp_excess[i] = out - set; }
    }
}

// Signatures and grouping keys of tables [t0, t0 + nt) for every probe.  Round 5: one THREAD per (probe, table) that
// walks the probe's k-mer hashes (the lanes of a wavefront cover two or three probes: the loads are broadcasts) and
// keeps the k minima in registers -- no wave reductions (the wavefront-per-probe form spent 450 LDS-routed shuffles per
// probe on its 75 wave-mins) and a 32 x 32 -> 64-bit multiply with the Mersenne fold 2^32 = 2 (mod p) instead of the
// 64 x 64-bit product the compiler made of `a * x + b` on u64 operands: 552 -> ~200 ms per S5 step.
// (The first version, one thread per probe and table walking its own row of xs with a load per function, fetched 64
// different cache lines per load instruction: 17 ms per table for 3.9 M probes; here a thread loads each hash once.)
__device__ __forceinline__ u32 mh_axb_mod(u32 a, u32 x, u32 b) {     // (a x + b) mod (2^31 - 1), a, x < p, b <= p
    const u64 prod = (u64)a * (u64)x;                               // < 2^62
    const u64 v = (prod & 0xffffffffull) + 2ull * (prod >> 32) + (u64)b;   // 2^32 = 2 (mod p); v < 2^33
    u32 t = (u32)(v & MH_P) + (u32)(v >> 31);                       // < 2^31 + 4
    return min(t, t - (u32)MH_P);                                   // (t - p wraps when t < p)
}
#define MH_KF 3      // hash functions per pass over the k-mers = the k of the reference (3); 4 per pass computed a fourth, unused minimum: a quarter of the launch's multiplies
__global__ void __launch_bounds__(256)
mh_keys_all_kernel(const u32 *__restrict__ xs, const u32 *__restrict__ koff, u32 n, const u64 *__restrict__ ab,
                   int k, int ntables, int t0, int nt, const u32 *__restrict__ grp, u32 *__restrict__ sig_all,
                   u64 *__restrict__ keys_all, u32 *__restrict__ sig0T = nullptr, u32 tstride = 0) {
    const u64 g64 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g64 >= (u64)n * (u32)nt) return;
    const u32 i = (u32)(g64 / (u32)nt), tt = (u32)(g64 - (u64)i * (u32)nt);
    const int t = t0 + (int)tt;
    const u32 k0 = koff[i], nk = koff[i + 1] - k0;   // <= MH_MAXK = 256
    const u32 gidx = grp ? grp[i] : 0u;
    const u64 *abt = ab + ((size_t)gidx * ntables + t) * k * 2;
    u64 h = 0xcbf29ce484222325ull;
    if (grp) h = (h ^ (u64)gidx) * 0x100000001b3ull;   // the group is part of the key
    for (int f0 = 0; f0 < k; f0 += MH_KF) {
        u32 a[MH_KF], b[MH_KF], best[MH_KF];
#pragma unroll
        for (int q = 0; q < MH_KF; ++q) {
            const int f = min(f0 + q, k - 1);
            a[q] = (u32)(abt[f * 2] % MH_P); b[q] = (u32)abt[f * 2 + 1];
            // (opaque: the compiler folds zext(trunc(urem)) back into the 64-bit remainder and then multiplies BOTH of its
            // words by x in the loop -- two v_mad_u64_u32 per function and k-mer instead of one)
            asm volatile("" : "+v"(a[q]));
            best[q] = 0xffffffffu;
        }
        for (u32 y = 0; y < nk; ++y) {
            const u32 x = xs[k0 + y];
#pragma unroll
            for (int q = 0; q < MH_KF; ++q) best[q] = min(best[q], mh_axb_mod(a[q], x, b[q]));
        }
#pragma unroll
        for (int q = 0; q < MH_KF; ++q) {
            const int f = f0 + q;
            if (f >= k) break;
            sig_all[((size_t)tt * n + i) * k + f] = best[q];
            // (the first value of every table's signature once more, a row per probe: what owned_earlier looks at)
            if (f == 0 && sig0T) sig0T[(size_t)i * tstride + t] = best[q];
            h = (h ^ (u64)best[q]) * 0x100000001b3ull;
        }
    }
    keys_all[(size_t)tt * n + i] = h;
}

// the same for all tables at once: keys_all[t][i] -> (key, i) at t * n + i
__global__ void __launch_bounds__(256)
mh_tables_kernel(const u64 *__restrict__ keys_all, u32 n, u64 tn, u64 *__restrict__ keys, u32 *__restrict__ vals, int key_shift) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tn) return;
    keys[e] = keys_all[e] >> key_shift;
    vals[e] = (u32)(e % n);
}

__global__ void __launch_bounds__(256)
mh_table_kernel(const u64 *__restrict__ keys_all, u32 n, u64 *__restrict__ keys, u32 *__restrict__ vals, int key_shift = 0) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = keys_all[i] >> key_shift;
    vals[i] = i;
}

// exact Jaccard distance of two sorted lists of distinct k-mers <= thres ?
//   1.0 - float(len(a & b)) / len(a | b) <= thres   (near_duplicate_filter.py:155-157)
// The float64 expression is non-increasing in the intersection size (correctly
// rounded division and subtraction are monotone), so the smallest passing size
// is found first (a few evaluations of the exact expression) and the merge walk
// -- branch-free, a divergent three-way comparison would run all three paths --
// stops as soon as that size is reached or has become unreachable.  (S3-sized
// input, buckets of hundreds of near-identical probes: the plain walk took
// 84 ms per table, 98 % of the filter.)
__device__ __forceinline__ bool mh_near(const u64 *__restrict__ ah, const u64 *__restrict__ al, u32 na,
                                        const u64 *__restrict__ bh, const u64 *__restrict__ bl, u32 nb,
                                        const u32 *__restrict__ need_tab, const unsigned long long *__restrict__ fa,
                                        const unsigned long long *__restrict__ fb, u32 exa, u32 exb) {
    // need_tab[|A| + |B|] = the smallest intersection m with 1 - m / (|A| + |B| - m) <= thres (the reference's two IEEE
    // operations, evaluated on the host; what a binary search over m with a double division per step found here)
    const u32 most = min(na, nb);
    const u32 need = min(need_tab[na + nb], most + 1u);
    if (need > most) return false;
    if (need == 0) return true;
    u32 bound = min(exa, exb);
#pragma unroll 8
    for (int w = 0; w < MH_FPW; ++w) bound += (u32)__popcll(fa[w] & fb[w]);
    if (bound < need) return false;
    u32 x = 0, y = 0, inter = 0;
    while (x < na && y < nb) {
        const u64 h1 = ah[x], l1 = al[x], h2 = bh[y], l2 = bl[y];
        const u32 eq = (h1 == h2) & (l1 == l2);
        const u32 lt = (h1 < h2) | ((h1 == h2) & (l1 < l2));
        inter += eq;
        x += eq | lt;
        y += eq | (lt ^ 1u);
        if (inter >= need) return true;
        if (inter + min(na - x, nb - y) < need) return false;
    }
    return false;
}

// the same on the probes' sorted 2-bit codes (round 5: both probes of A/C/G/T only)
__device__ __forceinline__ bool mh_near_codes(const u32 *__restrict__ a, u32 na, const u32 *__restrict__ b, u32 nb,
                                              const u32 *__restrict__ need_tab, const unsigned long long *__restrict__ fa,
                                              const unsigned long long *__restrict__ fb, u32 exa, u32 exb) {
    const u32 most = min(na, nb);
    const u32 need = min(need_tab[na + nb], most + 1u);
    if (need > most) return false;
    if (need == 0) return true;
    u32 bound = min(exa, exb);
#pragma unroll 8
    for (int w = 0; w < MH_FPW; ++w) bound += (u32)__popcll(fa[w] & fb[w]);
    if (bound < need) return false;
    u32 x = 0, y = 0, inter = 0;
    while (x < na && y < nb) {
        const u32 c1 = a[x], c2 = b[y];
        const u32 eq = c1 == c2, lt = c1 < c2;
        inter += eq;
        x += eq | lt;
        y += eq | (lt ^ 1u);
        if (inter >= need) return true;
        if (inter + min(na - x, nb - y) < need) return false;
    }
    return false;
}

__global__ void __launch_bounds__(256)
mh_edge_kernel(const u32 *__restrict__ koff, const u32 *__restrict__ nuniq, const u64 *__restrict__ id_hi,
               const u64 *__restrict__ id_lo, const u32 *__restrict__ sig, int k, const u32 *__restrict__ need_tab,
               const unsigned long long *__restrict__ fp, const u32 *__restrict__ fp_excess, u32 n,
               const u64 *__restrict__ keys, const u32 *__restrict__ vals, const u32 *__restrict__ grp,
               u32 *__restrict__ e_i, u32 *__restrict__ e_j, u32 *__restrict__ count, u32 cap,
               const u32 *__restrict__ sig_earlier, int n_earlier) {
    const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const u64 key = keys[x];
    const u32 i = vals[x];
    u32 npairs = 0;
    for (u32 y = x; y-- > 0;) {
        if (keys[y] != key) break;
        const u32 j = vals[y];  // j < i: stable sort keeps indices ascending in a run
        bool same = !grp || grp[i] == grp[j];   // same bucket = same group and signature (the key only groups)
        for (int f = 0; f < k; ++f) same = same && sig[(size_t)i * k + f] == sig[(size_t)j * k + f];
        if (!same) continue;
        // The pair shared a bucket in an earlier table already: it was compared there and its edge
        // (if any) is in the list -- near-identical probes meet in almost every table, and their
        // distance is the same each time (S5 x 0.25: 13.2 s of the filter's 14 s were these repeats).
        bool seen = false;
        for (int tp = 0; tp < n_earlier && !seen; ++tp) {
            const u32 *si = sig_earlier + ((size_t)tp * n + i) * k, *sj = sig_earlier + ((size_t)tp * n + j) * k;
            bool eq = true;
            for (int f = 0; f < k; ++f) eq = eq && si[f] == sj[f];
            seen = eq;
        }
        if (seen) continue;
        ++npairs;
        if (mh_near(id_hi + koff[i], id_lo + koff[i], nuniq[i], id_hi + koff[j], id_lo + koff[j], nuniq[j],
                    need_tab, fp + (size_t)i * MH_FPW, fp + (size_t)j * MH_FPW, fp_excess[i], fp_excess[j])) {
            const u32 shard = (x >> 6) & (ES_SHARDS - 1);
            const u32 slot = atomicAdd(&count[shard * ES_STRIDE], 1u);
            if (slot < cap) { e_i[(size_t)shard * cap + slot] = i; e_j[(size_t)shard * cap + slot] = j; }
        }
    }
    if (npairs) atomicAdd((unsigned long long *)(count + ((x >> 6) & (ES_SHARDS - 1)) * ES_STRIDE + 2), (unsigned long long)npairs);
}

struct MinHashFamily {       // same signature in table t (the key only groups), exact Jaccard distance of the k-mer sets
    const u32 *koff, *nuniq;
    const u64 *id_hi, *id_lo;
    const u32 *sig_all;      // [table][probe][k]
    int k;
    u32 n;
    const u32 *grp;
    const u32 *need_tab;     // by |A| + |B|: the smallest intersection for which 1 - m / (|A| + |B| - m) <= thres
    const unsigned long long *fp;     // MH_FPW words per probe
    const u32 *fp_excess;
    int wave_near_max;       // lanes wanting a comparison from which every lane runs its own (NDF_WAVE_NEAR_MAX)
    const u32 *kc;           // [probe][kstride] 2-bit codes of the distinct k-mers, MH_KC_NONE-padded (row[0] == NONE: not packable)
    u32 kstride;
    const u32 *sig0T;        // [probe][tstride]: the first value of every table's signature
    u32 tstride;
    static constexpr bool WAVE_NEAR = true;
    static constexpr bool WAVE64 = true;
    // (round 6: the two forms of a wavefront's scratch are never live together -- near_wave stages a MATE's ids / codes per
    // comparison, the shared walk stages the WALKING probe's table for the length of its loop -- so they share their 4 KB:
    // 16 instead of 32 KB of LDS per workgroup, which was what held ndf_probe_kernel at 5 wavefronts per SIMD)
    struct Scratch { union { struct { u64 h[MH_MAXK], l[MH_MAXK]; }; u32 tab[MH_TAB]; }; };
    // The shared walk (round 5).  A wavefront that walks ONE probe's run compares that probe with thousands of mates:
    // its k-mer codes go into an open-addressing table in LDS once (wave64_stage), and a comparison is then every lane
    // looking ITS k-mers of the mate up there -- the mate's row read coalesced, no dependent global loads, ~40
    // instructions per comparison for the wavefront, four mates in flight -- instead of a merge walk per lane
    // (~100 dependent steps with their global loads, a full step of 64 mates ~50 us).
    __device__ __forceinline__ u32 tab_slot(u32 c) const { return (c * 0x9E3779B1u) >> (32 - MH_TAB_BITS); }
    __device__ __forceinline__ bool wave64_stage(u32 i, Scratch &S, u32 lane) const {      // i wave-uniform
        if (!kc || kc[(size_t)i * kstride] == MH_KC_NONE) return false;
        for (u32 q = lane; q < MH_TAB; q += 64) S.tab[q] = MH_KC_NONE;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const u32 na = nuniq[i];
        for (u32 q = lane; q < na; q += 64) {
            const u32 c = kc[(size_t)i * kstride + q];
            u32 sl = tab_slot(c);
            while (atomicCAS(&S.tab[sl], MH_KC_NONE, c) != MH_KC_NONE) sl = (sl + 1u) & (MH_TAB - 1u);   // (codes are distinct)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return true;
    }
    __device__ __forceinline__ u32 tab_has(const Scratch &S, u32 c) const {
        if (c == MH_KC_NONE) return 0u;
        u32 sl = tab_slot(c);
        for (;;) {
            const u32 v = S.tab[sl];
            if (v == c) return 1u;
            if (v == MH_KC_NONE) return 0u;
            sl = (sl + 1u) & (MH_TAB - 1u);
        }
    }
    // the first mate near i among the lanes in `want` (lane b holds mate j), in lane order; -1: none.  The staged probe
    // is i; ncmp counts the comparisons made (the walk ends at the first near mate, so the ones after it are not made
    // -- up to three of its batch are).
    __device__ __forceinline__ int wave64_first_near(u32 i, unsigned long long want, u32 j, Scratch &S, u32 lane, u32 &ncmp) const {
        const u32 na = __builtin_amdgcn_readfirstlane(nuniq[i]);
        while (want) {
            int b[4];
            u32 jb[4], nbv[4], need[4], cnt[4];
            bool slow[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                b[m] = want ? __ffsll((long long)want) - 1 : -1;
                if (want) want &= want - 1ull;
                jb[m] = (u32)__builtin_amdgcn_readfirstlane(__shfl((int)j, b[m] < 0 ? 0 : b[m], WAVE));
                nbv[m] = 0; need[m] = 1; cnt[m] = 0; slow[m] = false;
                if (b[m] >= 0) {
                    nbv[m] = __builtin_amdgcn_readfirstlane(nuniq[jb[m]]);
                    need[m] = min(need_tab[na + nbv[m]], min(na, nbv[m]) + 1u);
                    slow[m] = kc[(size_t)jb[m] * kstride] == MH_KC_NONE;     // (a mate with other characters: the merge walk, by one lane)
                }
            }
            ncmp += (lane == 0) ? (u32)((b[0] >= 0) + (b[1] >= 0) + (b[2] >= 0) + (b[3] >= 0)) : 0u;
            const u32 nbmax = max(max(nbv[0], nbv[1]), max(nbv[2], nbv[3]));
            for (u32 q0 = 0; q0 < nbmax; q0 += 64) {
                u32 c[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    c[m] = (b[m] >= 0 && !slow[m] && q0 + lane < nbv[m]) ? kc[(size_t)jb[m] * kstride + q0 + lane] : MH_KC_NONE;
#pragma unroll
                for (int m = 0; m < 4; ++m) cnt[m] += (u32)__popcll(__ballot(tab_has(S, c[m]) != 0u));
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (b[m] < 0) break;
                bool r;
                if (slow[m]) {
                    bool mine = false;
                    if ((int)lane == b[m]) mine = near(0, i, jb[m]);
                    r = __ballot(mine) != 0ull;
                } else {
                    r = need[m] <= min(na, nbv[m]) && cnt[m] >= need[m];
                }
                if (r) return b[m];
            }
        }
        return -1;
    }
    // i near j, by the whole wavefront (i, j wave-uniform): B's k-mers staged in LDS, every lane looks its share of
    // A's up by binary search, the matches are summed -- |A and B| >= need is what mh_near's walk decides (its early
    // exits only shorten the walk), `need` from a table of the reference's own two IEEE operations made on the host
    __device__ __forceinline__ bool near_wave(u32 i, u32 j, Scratch &S, u32 lane) const {
        const u32 na = nuniq[i], nb = nuniq[j];
        const u32 most = min(na, nb);
        const u32 need = min(need_tab[na + nb], most + 1u);
        if (need > most) return false;
        if (need == 0) return true;
        {   // the fingerprint bound, a word per lane
            u32 b = lane < MH_FPW ? (u32)__popcll(fp[(size_t)i * MH_FPW + lane] & fp[(size_t)j * MH_FPW + lane]) : 0u;
            for (int d = 32; d > 0; d >>= 1) b += __shfl_xor(b, d, WAVE);
            if (b + min(fp_excess[i], fp_excess[j]) < need) return false;
        }
        if (kc && kc[(size_t)i * kstride] != MH_KC_NONE && kc[(size_t)j * kstride] != MH_KC_NONE) {
            // both probes have their 2-bit codes (round 5): B's sorted codes in LDS, 32-bit binary searches
            u32 *sb = (u32 *)S.h;
            const u32 *bc = kc + (size_t)j * kstride, *ac = kc + (size_t)i * kstride;
            for (u32 q = lane; q < nb; q += 64) sb[q] = bc[q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            u32 cnt = 0;
            for (u32 q0 = 0; q0 < na; q0 += 64) {
                const u32 q = q0 + lane;
                const bool have = q < na;
                const u32 c = have ? ac[q] : 0u;
                u32 lo = 0, hi = nb;                    // first element of B that is not below c
                while (__ballot(lo < hi)) {
                    if (lo < hi) {
                        const u32 mid = (lo + hi) >> 1;
                        if (sb[mid] < c) lo = mid + 1; else hi = mid;
                    }
                }
                cnt += (have && lo < nb && sb[lo] == c) ? 1u : 0u;
            }
            for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, WAVE);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();            // (S is rewritten by the next comparison)
            return cnt >= need;
        }
        const u64 *bh = id_hi + koff[j], *bl = id_lo + koff[j];
        for (u32 q = lane; q < nb; q += 64) { S.h[q] = bh[q]; S.l[q] = bl[q]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const u64 *ah = id_hi + koff[i], *al = id_lo + koff[i];
        u32 cnt = 0;
        for (u32 q0 = 0; q0 < na; q0 += 64) {
            const u32 q = q0 + lane;
            const bool have = q < na;
            const u64 h = have ? ah[q] : 0ull, l = have ? al[q] : 0ull;
            u32 lo = 0, hi = nb;                        // first element of B that is not below (h, l)
            while (__ballot(lo < hi)) {
                if (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    const u64 mh_ = S.h[mid], ml_ = S.l[mid];
                    if (mh_ < h || (mh_ == h && ml_ < l)) lo = mid + 1; else hi = mid;
                }
            }
            cnt += (have && lo < nb && S.h[lo] == h && S.l[lo] == l) ? 1u : 0u;
        }
        for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, WAVE);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                // (S is rewritten by the next comparison)
        return cnt >= need;
    }
    __device__ __forceinline__ bool same_sig(u32 t, u32 i, u32 j) const {
        const u32 *si = sig_all + ((size_t)t * n + i) * k, *sj = sig_all + ((size_t)t * n + j) * k;
        bool eq = true;
        for (int f = 0; f < k; ++f) eq = eq && si[f] == sj[f];
        return eq;
    }
    __device__ __forceinline__ bool same_bucket(u32 t, u32 i, u32 j) const { return (!grp || grp[i] == grp[j]) && same_sig(t, i, j); }
    // (eight earlier tables per round trip: the first values of both signatures are requested together, a table whose
    // first values agree is then compared in full -- one table at a time was a chain of up to 24 dependent loads per
    // mate, and since the comparisons are shared by the wavefront the walks wait for loads, not for the ALU)
    __device__ __forceinline__ bool owned_earlier(u32 t, u32 i, u32 j) const {
        for (u32 t0 = 0; t0 < t; t0 += 8) {
            u32 a[8], b[8];
#pragma unroll
            for (u32 q = 0; q < 8; ++q) {
                const u32 tp = min(t0 + q, t - 1);
                a[q] = sig0T ? sig0T[(size_t)i * tstride + tp] : sig_all[((size_t)tp * n + i) * k];   // (a row per probe: one line
                b[q] = sig0T ? sig0T[(size_t)j * tstride + tp] : sig_all[((size_t)tp * n + j) * k];   //  instead of eight)
            }
#pragma unroll
            for (u32 q = 0; q < 8; ++q)
                if (t0 + q < t && a[q] == b[q] && same_sig(t0 + q, i, j)) return true;
        }
        return false;
    }
    __device__ __forceinline__ bool near(u32, u32 i, u32 j) const {
        if (kc && kc[(size_t)i * kstride] != MH_KC_NONE && kc[(size_t)j * kstride] != MH_KC_NONE)
            return mh_near_codes(kc + (size_t)i * kstride, nuniq[i], kc + (size_t)j * kstride, nuniq[j], need_tab,
                                 fp + (size_t)i * MH_FPW, fp + (size_t)j * MH_FPW, fp_excess[i], fp_excess[j]);
        return mh_near(id_hi + koff[i], id_lo + koff[i], nuniq[i], id_hi + koff[j], id_lo + koff[j], nuniq[j],
                       need_tab, fp + (size_t)i * MH_FPW, fp + (size_t)j * MH_FPW, fp_excess[i], fp_excess[j]);
    }
};

// off[i] = i * a for i <= n (the offsets of equal-length rows)
__global__ void __launch_bounds__(256)
ndf_iota_mul_kernel(u32 *__restrict__ out, u32 n_plus_1, u32 a) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_plus_1) out[i] = i * a;
}

// bytes_on_device: `bytes` already lives on the device (>= probe_off[n] + 16 bytes allocated)
// Equal-length rows on the device (round 5: the candidates object's calls; the host arrays of 5 M offsets, k-mer
// offsets and groups, their loops and their pageable uploads were ~100 ms per call beside ~130 ms of kernels):
// probe_off == nullptr and equal_len = the row length; the group of every row as a device array (d_grp_dev, or none);
// the verdicts as flags on the device (keep.d_flags).
static int ndf_minhash_impl(catchhip_ctx *ctx, const u8 *bytes, const i64 *probe_off, i64 n, const i64 *group_off,
                            i64 ngroups, i32 kmer_size, const i64 *ab, i32 ntables, i32 k, double dist_thres,
                            NdfKeep keep, bool bytes_on_device = false, i64 equal_len = 0, const u32 *d_grp_dev = nullptr) {
    ARG_CHECK(ctx && n >= 0 && kmer_size >= 1 && kmer_size <= 16 && ntables >= 1 && k >= 1 && k <= 16 && ab);
    PoolScope pool_scope(ctx);
    if (n == 0) return 0;
    ARG_CHECK(bytes && (keep.host || keep.d_flags));
    ARG_CHECK(probe_off ? probe_off[0] == 0 : (bytes_on_device && equal_len > 0));
    ARG_CHECK(!(group_off && d_grp_dev));
    std::vector<u32> h_grp;
    if (group_off) {
        ARG_CHECK(ngroups >= 1 && group_off[0] == 0 && group_off[ngroups] == n);
        h_grp.resize((size_t)n);
        for (i64 g = 0; g < ngroups; ++g) {
            ARG_CHECK(group_off[g + 1] >= group_off[g]);
            for (i64 i = group_off[g]; i < group_off[g + 1]; ++i) h_grp[(size_t)i] = (u32)g;
        }
    } else if (d_grp_dev) {
        ARG_CHECK(ngroups >= 1);
    } else {
        ngroups = 1;
    }
    const i64 total_bytes = probe_off ? probe_off[n] : n * equal_len;
    ARG_CHECK(n < ((i64)1 << 31) && total_bytes < ((i64)1 << 32));
    std::vector<u32> h_off, h_koff;
    u32 max_nk = 1;
    size_t nkm = 0;
    if (probe_off) {
        h_off.resize((size_t)n + 1);
        h_koff.assign((size_t)n + 1, 0);
        for (i64 i = 0; i <= n; ++i) h_off[i] = (u32)probe_off[i];
        for (i64 i = 0; i < n; ++i) {
            const i64 len = probe_off[i + 1] - probe_off[i];
            // lsh.py:113 asserts kmer_size <= len(s)
            if (len < kmer_size) { chip_set_error("ndf minhash: probe %lld shorter than the k-mer size", (long long)i); return CATCHHIP_EINVAL; }
            const i64 nk = len - kmer_size + 1;
            if (nk > MH_MAXK) { chip_set_error("ndf minhash: more than %d k-mers per probe not supported", MH_MAXK); return CATCHHIP_EINVAL; }
            h_koff[i + 1] = h_koff[i] + (u32)nk;
            max_nk = std::max(max_nk, (u32)nk);
        }
        nkm = h_koff[n];
    } else {
        if (equal_len < kmer_size) { chip_set_error("ndf minhash: probes shorter than the k-mer size"); return CATCHHIP_EINVAL; }
        const i64 nk = equal_len - kmer_size + 1;
        if (nk > MH_MAXK) { chip_set_error("ndf minhash: more than %d k-mers per probe not supported", MH_MAXK); return CATCHHIP_EINVAL; }
        ARG_CHECK(n * nk < ((i64)1 << 32));
        max_nk = (u32)nk;
        nkm = (size_t)(n * nk);
    }
    for (i64 t = 0; t < ngroups * (i64)ntables * k; ++t)
        ARG_CHECK(ab[2 * t] >= 1 && ab[2 * t] <= (i64)MH_P && ab[2 * t + 1] >= 0 && ab[2 * t + 1] <= (i64)MH_P);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const u32 nn = (u32)n;
    const size_t total = (size_t)total_bytes;
    DevBuf<u8> d_bytes_own;
    DevBuf<u64> d_ab, keys, keys_alt, id_hi, id_lo;
    DevBuf<u32> d_off, d_koff, xs, nuniq, sig, vals, vals_alt, e_i, e_j, count, status, flags, d_grp;
    if (!bytes_on_device) TRY(d_bytes_own.alloc(total + 16));
    struct { const u8 *p; } d_bytes = {bytes_on_device ? bytes : (const u8 *)d_bytes_own.p};
    TRY(d_ab.alloc((size_t)ngroups * ntables * k * 2));
    if (group_off) TRY(d_grp.alloc((size_t)n));
    TRY(d_off.alloc((size_t)n + 1));
    TRY(d_koff.alloc((size_t)n + 1));
    TRY(xs.alloc(nkm));
    DevBuf<u64> fp;
    DevBuf<u32> fp_excess, need_tab;
    TRY(fp.alloc((size_t)nn * MH_FPW));
    TRY(fp_excess.alloc(nn));
    TRY(need_tab.alloc(2 * MH_MAXK + 2));
    {
        // the intersection the threshold needs, by |A| + |B| (mh_near, MinHashFamily::near_wave): the reference's
        // expression, evaluated here with the same two IEEE operations
        std::vector<u32> h_need(2 * MH_MAXK + 2);
        for (u32 S = 0; S < h_need.size(); ++S) {
            u32 m_ = 0;
            for (; m_ <= S / 2; ++m_) {
                volatile double sim = (double)m_ / (double)(S - m_);
                volatile double dist = 1.0 - sim;
                if (dist <= dist_thres) break;
            }
            h_need[S] = m_;
        }
        TRY(chip_pinned_reserve(ctx, sizeof(u32) * h_need.size()));
        memcpy(ctx->h_big, h_need.data(), sizeof(u32) * h_need.size());
        HIP_TRY(hipMemcpyAsync(need_tab.p, ctx->h_big, sizeof(u32) * h_need.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));           // (h_big is reused below)
    }
    TRY(nuniq.alloc(nn));
    // signatures / keys of a chunk of tables at a time (<= 16 GB of signatures: all tables of any input
    // that fits a probes object, so that a pair is compared in the first table it meets in only)
    const int tchunk = (int)std::max<i64>(1, std::min<i64>(ntables, ((i64)1 << 32) / std::max<i64>(1, n * k)));
    DevBuf<u64> keys_all;
    TRY(sig.alloc((size_t)tchunk * nn * k));
    TRY(keys_all.alloc((size_t)tchunk * nn));
    TRY(keys.alloc(nn));
    TRY(vals.alloc(nn));
    TRY(count.alloc(ES_WORDS + 8));
    TRY(status.alloc(nn));
    TRY(flags.alloc(nn));
    if (!bytes_on_device) HIP_TRY(hipMemcpyAsync(d_bytes_own.p, bytes, total, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_ab.p, ab, sizeof(i64) * (size_t)ngroups * ntables * k * 2, hipMemcpyHostToDevice, s));
    if (group_off) HIP_TRY(hipMemcpyAsync(d_grp.p, h_grp.data(), sizeof(u32) * (size_t)n, hipMemcpyHostToDevice, s));
    const u32 *grp = group_off ? (const u32 *)d_grp.p : d_grp_dev;
    if (probe_off) {
        HIP_TRY(hipMemcpyAsync(d_off.p, h_off.data(), sizeof(u32) * (n + 1), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_koff.p, h_koff.data(), sizeof(u32) * (n + 1), hipMemcpyHostToDevice, s));
    } else {
        hipLaunchKernelGGL(ndf_iota_mul_kernel, dim3((unsigned)div_up(n + 1, 256)), dim3(256), 0, s, d_off.p, (u32)n + 1u, (u32)equal_len);
        hipLaunchKernelGGL(ndf_iota_mul_kernel, dim3((unsigned)div_up(n + 1, 256)), dim3(256), 0, s, d_koff.p, (u32)n + 1u, max_nk);
    }
    HIP_TRY(hipMemsetAsync(count.p, 0, ES_WORDS * sizeof(u32), s));
    HIP_TRY(hipMemsetAsync(status.p, 0, sizeof(u32) * nn, s));
    HIP_TRY(hipMemsetAsync(flags.p, 0, sizeof(u32) * nn, s));

    PhaseTimer tm(ctx, PHASE_NDF);
    const unsigned nb = (unsigned)div_up(nn, 256);
    const bool lazy = tchunk >= ntables && (i64)ntables * n <= ((i64)1 << 31) && !chip_test_env("CATCHHIP_MH_ALL_PAIRS") &&
                      ndf_lazy_fits((size_t)ntables * nn);
    // the shared walks' packed k-mer codes and the transposed first signature values (lazy resolution only;
    // CATCHHIP_MH_NO_WAVE64, a test hook, keeps round 4's walks for the equality tests)
    const bool wave64 = lazy && !chip_test_env("CATCHHIP_MH_NO_WAVE64");
    DevBuf<u32> kc, sig0T;
    const u32 kstride = (max_nk + 31u) & ~31u, tstride = ((u32)ntables + 3u) & ~3u;
    if (wave64) TRY(kc.alloc((size_t)nn * kstride));
    if (lazy) TRY(sig0T.alloc((size_t)nn * tstride));
    // The 16-byte k-mer ids only where a probe may lack its codes: any character outside A/C/G/T, k-mers of 16, or no
    // code rows at all (the edge-list variant)
    bool want_ids = !wave64 || kmer_size > 15;
    if (!want_ids) {
        HIP_TRY(hipMemsetAsync(count.p, 0, sizeof(u32), s));
        hipLaunchKernelGGL(mh_other_chars_kernel, dim3(2048), dim3(256), 0, s, (const u8 *)d_bytes.p, (u64)total, count.p);
        HIP_TRY(hipMemcpyAsync(ctx->h_pin, count.p, sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        want_ids = *(volatile u32 *)ctx->h_pin != 0u;
        HIP_TRY(hipMemsetAsync(count.p, 0, sizeof(u32), s));
    }
    if (want_ids) { TRY(id_hi.alloc(nkm)); TRY(id_lo.alloc(nkm)); }
    hipLaunchKernelGGL(kmer_size == 10 ? mh_kmer_kernel<10> : mh_kmer_kernel<0>, dim3((unsigned)std::min<i64>(n, (i64)1 << 20)), dim3(64), 0, s,
                       (const u8 *)d_bytes.p, (const u32 *)d_off.p, (const u32 *)d_koff.p, nn, (int)kmer_size, xs.p,
                       id_hi.p, id_lo.p, nuniq.p, (unsigned long long *)fp.p, fp_excess.p, wave64 ? kc.p : (u32 *)nullptr, kstride);
    tm.launch(1);
    ctx->ndf_counters[0] = n; ctx->ndf_counters[1] = ntables; ctx->ndf_counters[2] = ctx->ndf_counters[3] = 0;
    if (lazy) {
        // lazy resolution (ndf_lazy_kernel): all tables' sorted runs stay resident, one cursor per (table, slot) --
        // 24 bytes per entry, at most 2^31 entries (48 GB); beyond that the edge-list variant below, table by table
        DevBuf<u64> skeys, pairs;
        DevBuf<u32> svals, cursor;
        const size_t tn = (size_t)ntables * nn;
        const bool timing = getenv("CATCHHIP_TIMING") != nullptr;
        auto lap = [&](const char *what, std::chrono::steady_clock::time_point &t0) {
            if (!timing) return;
            (void)hipStreamSynchronize(s);
            const auto t1 = std::chrono::steady_clock::now();
            fprintf(stderr, "[catchhip]   minhash filter, %u probes, %d tables: %s %.1f ms\n", nn, (int)ntables, what,
                    std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        };
        auto t_lap = std::chrono::steady_clock::now();
        lap("k-mers", t_lap);
        TRY(skeys.alloc(tn));
        TRY(svals.alloc(tn));
        const u32 TS = ndf_table_stride((u32)ntables);
        TRY(cursor.alloc((size_t)nn * TS));
        TRY(pairs.alloc(2 * ES_SHARDS));
        HIP_TRY(hipMemsetAsync(cursor.p, 0xff, sizeof(u32) * (size_t)nn * TS, s));
        HIP_TRY(hipMemsetAsync(pairs.p, 0, sizeof(u64) * 2 * ES_SHARDS, s));
        hipLaunchKernelGGL(mh_keys_all_kernel, dim3((unsigned)div_up((i64)nn * ntables, 256)), dim3(256), 0, s,
                           (const u32 *)xs.p, (const u32 *)d_koff.p, nn, (const u64 *)d_ab.p, (int)k, (int)ntables, 0,
                           (int)ntables, grp, sig.p, keys_all.p, sig0T.p, tstride);
        tm.launch(1);
        if (chip_test_env("CATCHHIP_MH_SORT_ONE_BY_ONE")) {      // (test hook: round 4's table-by-table sorts)
            for (int t = 0; t < ntables; ++t) {
                hipLaunchKernelGGL(mh_table_kernel, dim3(nb), dim3(256), 0, s, (const u64 *)(keys_all.p + (size_t)t * nn), nn,
                                   keys.p, vals.p, 32);
                TRY(chip_radix_sort_pairs(ctx, keys, keys_alt, vals, vals_alt, nn, 32));
                HIP_TRY(hipMemcpyAsync(skeys.p + (size_t)t * nn, keys.p, sizeof(u64) * nn, hipMemcpyDeviceToDevice, s));
                HIP_TRY(hipMemcpyAsync(svals.p + (size_t)t * nn, vals.p, sizeof(u32) * nn, hipMemcpyDeviceToDevice, s));
                tm.launch(1 + 24 + 2);
            }
        } else {
            // all tables' (key, probe) pairs side by side, sorted as segments by one set of launches
            DevBuf<u64> skeys_alt;
            DevBuf<u32> svals_alt;
            hipLaunchKernelGGL(mh_tables_kernel, dim3((unsigned)div_up((i64)tn, 256)), dim3(256), 0, s, (const u64 *)keys_all.p, nn,
                               (u64)tn, skeys.p, svals.p, 32);
            TRY(chip_radix_sort_pairs_segments(ctx, skeys, skeys_alt, svals, svals_alt, nn, ntables, 32));
            tm.launch(1 + 24);
        }
        lap("signatures + sorts", t_lap);
        MinHashFamily fam{(const u32 *)d_koff.p, (const u32 *)nuniq.p, (const u64 *)id_hi.p, (const u64 *)id_lo.p, (const u32 *)sig.p,
                          (int)k, nn, grp, (const u32 *)need_tab.p, (const unsigned long long *)fp.p, (const u32 *)fp_excess.p,
                          chip_test_env("CATCHHIP_NDF_WAVE_NEAR_MAX") ? atoi(chip_test_env("CATCHHIP_NDF_WAVE_NEAR_MAX")) : NDF_WAVE_NEAR_MAX,
                          (const u32 *)kc.p, kstride, (const u32 *)sig0T.p, tstride};
        const int rc = ndf_lazy_rounds(ctx, nn, tn, count, status, flags, pairs, tm, keep,
                                       [&](bool wake, const u32 *list, u32 nlist, u32 *next, u32 *next_count, u32 *left_tables, const u32 *inv, u32 nslot, NdfQueue Q, int mode) {
            if (wake && mode == 2)
                hipLaunchKernelGGL((ndf_probe_kernel<MinHashFamily>), dim3((unsigned)div_up((i64)nlist * 64, 256)), dim3(256), 0, s,
                                   fam, nn, (const u64 *)skeys.p, (const u32 *)svals.p, cursor.p, status.p, flags.p,
                                   (unsigned long long *)pairs.p, list, nlist, left_tables, inv, (u32)ntables, nslot, TS, wave64, Q);
            else if (wake && mode == 1)
                hipLaunchKernelGGL((ndf_lazy_kernel<MinHashFamily, true, true>), dim3(mode == 1 ? NDF_DRAIN_BLOCKS : (unsigned)div_up(list ? (i64)nlist * ntables : (i64)nlist, 256)), dim3(256), 0, s,
                                   fam, nn, (const u64 *)skeys.p, (const u32 *)svals.p, cursor.p, status.p, flags.p,
                                   (unsigned long long *)pairs.p, list, nlist, next, next_count, left_tables, inv, (u32)ntables, nslot, TS, wave64, Q);
                        else if (wake)
                hipLaunchKernelGGL((ndf_lazy_kernel<MinHashFamily, true>), dim3(mode == 1 ? NDF_DRAIN_BLOCKS : (unsigned)div_up(list ? (i64)nlist * ntables : (i64)nlist, 256)), dim3(256), 0, s,
                                   fam, nn, (const u64 *)skeys.p, (const u32 *)svals.p, cursor.p, status.p, flags.p,
                                   (unsigned long long *)pairs.p, list, nlist, next, next_count, left_tables, inv, (u32)ntables, nslot, TS, wave64, Q);
            else
                hipLaunchKernelGGL((ndf_lazy_kernel<MinHashFamily, false>), dim3((unsigned)div_up((i64)nlist, 256)), dim3(256), 0, s, fam, nn,
                                   (const u64 *)skeys.p, (const u32 *)svals.p, cursor.p, status.p, flags.p,
                                   (unsigned long long *)pairs.p, list, nlist, next, next_count, left_tables, inv, (u32)ntables, nslot, TS, wave64, Q);
        }, skeys, svals, cursor, (u32)ntables);
        lap("rounds", t_lap);
        return rc;
    }
    u32 cap = (u32)std::max<i64>((i64)1 << 14, std::min<i64>(n * 16, (i64)1 << 28) / ES_SHARDS);   // per shard
    u32 ne = 0;
    for (int attempt = 0;; ++attempt) {
        TRY(e_i.reserve((size_t)cap * ES_SHARDS));
        TRY(e_j.reserve((size_t)cap * ES_SHARDS));
        HIP_TRY(hipMemsetAsync(count.p, 0, ES_WORDS * sizeof(u32), s));
        for (int t = 0; t < ntables; ++t) {
            const int tc = t % tchunk;
            if (tc == 0) {
                hipLaunchKernelGGL(mh_keys_all_kernel, dim3((unsigned)div_up((i64)nn * std::min(tchunk, ntables - t), 256)), dim3(256), 0, s,
                                   (const u32 *)xs.p, (const u32 *)d_koff.p, nn, (const u64 *)d_ab.p, (int)k,
                                   (int)ntables, t, std::min(tchunk, ntables - t), grp, sig.p, keys_all.p);
                tm.launch(1);
            }
            hipLaunchKernelGGL(mh_table_kernel, dim3(nb), dim3(256), 0, s,
                               (const u64 *)(keys_all.p + (size_t)tc * nn), nn, keys.p, vals.p);
            TRY(chip_radix_sort_pairs(ctx, keys, keys_alt, vals, vals_alt, nn, 64));
            hipLaunchKernelGGL(mh_edge_kernel, dim3(nb), dim3(256), 0, s, (const u32 *)d_koff.p, (const u32 *)nuniq.p,
                               (const u64 *)id_hi.p, (const u64 *)id_lo.p,
                               (const u32 *)(sig.p + (size_t)tc * nn * k), (int)k, (const u32 *)need_tab.p,
                               (const unsigned long long *)fp.p, (const u32 *)fp_excess.p, nn,
                               (const u64 *)keys.p, (const u32 *)vals.p, grp, e_i.p, e_j.p, count.p, cap,
                               (const u32 *)sig.p, chip_test_env("CATCHHIP_MH_NO_DEDUPE") ? 0 : tc);
            tm.launch(2 + 24);
        }
        HIP_TRY(hipGetLastError());
        TRY(ndf_fullest_shard(ctx, count.p, &ne));
        if (ne <= cap) break;
        if (attempt >= 2) { chip_set_error("ndf: edge buffer overflow"); return CATCHHIP_ENOMEM; }
        cap = ne;
    }
    return ndf_resolve(ctx, nn, ne, cap, e_i, e_j, count, status, flags, tm, keep);
}

extern "C" int catchhip_ndf_minhash(catchhip_ctx *ctx, const u8 *bytes, const i64 *probe_off, i64 n, i32 kmer_size,
                                    const i64 *ab, i32 ntables, i32 k, double dist_thres, u8 *keep) {
    ARG_CHECK(keep || n == 0);
    return ndf_minhash_impl(ctx, bytes, probe_off, n, nullptr, 1, kmer_size, ab, ntables, k, dist_thres, NdfKeep{keep, nullptr});
}

extern "C" int catchhip_ndf_minhash_many(catchhip_ctx *ctx, const u8 *bytes, const i64 *probe_off, i64 n,
                                         const i64 *group_off, i64 ngroups, i32 kmer_size, const i64 *ab,
                                         i32 ntables, i32 k, double dist_thres, u8 *keep) {
    ARG_CHECK(group_off && ngroups >= 1);
    ARG_CHECK(keep || n == 0);
    return ndf_minhash_impl(ctx, bytes, probe_off, n, group_off, ngroups, kmer_size, ab, ntables, k, dist_thres,
                            NdfKeep{keep, nullptr});
}

// the MinHash filter on probes whose characters are already on the device
int chip_ndf_minhash_device(catchhip_ctx *ctx, const u8 *d_rows, const i64 *probe_off, i64 n, const i64 *group_off,
                            i64 ngroups, i32 kmer_size, const i64 *ab, i32 ntables, i32 k, double dist_thres,
                            u8 *keep) {
    ARG_CHECK(keep || n == 0);
    return ndf_minhash_impl(ctx, d_rows, probe_off, n, group_off, ngroups, kmer_size, ab, ntables, k, dist_thres,
                            NdfKeep{keep, nullptr}, true);
}
// the same on n rows of L characters, groups (if any) and verdicts on the device: d_keep_flags[i] = 1 kept, 0 dropped
int chip_ndf_minhash_rows(catchhip_ctx *ctx, const u8 *d_rows, i64 n, i64 L, const u32 *d_grp, i64 ngroups, i32 kmer_size,
                          const i64 *ab, i32 ntables, i32 k, double dist_thres, u32 *d_keep_flags) {
    ARG_CHECK(d_keep_flags || n == 0);
    return ndf_minhash_impl(ctx, d_rows, nullptr, n, nullptr, d_grp ? ngroups : 1, kmer_size, ab, ntables, k, dist_thres,
                            NdfKeep{nullptr, d_keep_flags}, true, L, d_grp);
}
