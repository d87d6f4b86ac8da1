// K1: probe -> target coverage scan.
//
// Three exact ways to find the (probe, offset) pairs that the reference's
// find_probe_covers_in_sequence reports (catch/probe.py:1008-1271):
//  * seed scan (K1c, default): valid when every probe has the same length L,
//    anchors are the pigeonhole anchors {0,k,..,L-k} with L/k > mismatches,
//    lcf_thres == L, island == 0, the alphabet is A/C/G/T/N and every target
//    sequence is at least L long: then a probe covers
//    offset o iff Hamming(probe, seq[o:o+L]) <= mismatches (character
//    equality, N == N) and the cover range is exactly (o, o+L).  A hash table
//    of the anchor k-mers seeds the pairs, one thread per seed verifies on the
//    3 bit-planes (32 bases per word, XOR/OR + v_bcnt).
//  * tiled scan (scan_fast3_kernel, CATCHHIP_SCAN_FAST): same conditions,
//    every probe against every offset with a 32-base lower-bound filter; the
//    O(P*G) cross-check of the seed scan.
//  * general path (any alphabet, any anchors, truncated alignments,
//    lcf_thres < L, island): seed join (hash every target k-mer, binary
//    search in the sorted anchor-hash table) + one lane per seed hit that
//    evaluates the reference's cover function on the raw bytes.
// All three hand their hits to the bucketed row build (rows_bucket.inc).
#include <algorithm>
#include <chrono>

#include "internal.h"

// ------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------
// index s with off[s] <= x < off[s+1]  (off has n+1 entries, off[0] = 0)
__device__ __forceinline__ u32 find_segment(const u32 *__restrict__ off, u32 n, u32 x) {
    u32 lo = 0, hi = n;  // answer in [lo, hi)
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (off[mid] <= x) lo = mid; else hi = mid;
    }
    // skip empty segments: off[lo] <= x and we need x < off[lo+1]
    while (lo + 1 < n && off[lo + 1] <= x) ++lo;
    return lo;
}

struct HitBuf {
    u32 *a;      // probe (unique index)
    u32 *b;      // global start
    u32 *c;      // global end (general path) -- may be null in fast path
    u32 *count;  // device counter
    u32 cap;
    u32 *d;      // optional (general path): global position of the seeding k-mer
    u32 *e;      // optional (general path): anchor entry of the seed
};

// ------------------------------------------------------------------------
// tiled scan: filter on the first 32 bases, verify the rest.
//
// The hot loop only needs a LOWER bound of the mismatch count to reject a
// (probe, offset) pair: the XOR/OR of planes 0 and 1 over the first 32 bases
// never over-counts (plane 2 -- "is not A/C/G/T" -- and the remaining words can
// only add mismatches).  So each lane keeps just word 0 of planes 0/1 for
// SF2_OPL = 16 offsets (32 VGPRs), a probe's word 0 arrives through the scalar
// cache (wave-uniform address -> s_load into SGPRs, used directly as the scalar
// operand), and per pair the loop issues v_xor, v_bitop3, v_bcnt, 1/2 v_min3.
// When some lane of the wave stays within the budget (rare: only near true
// homology) the wave verifies those offsets exactly, reading the target
// windows and the probe's full bit-planes from global memory (L2-resident).
// ------------------------------------------------------------------------
#define SF2_THREADS 256
#define SF2_OPL 16
#define SF2_TILE (SF2_THREADS * SF2_OPL)

__device__ __forceinline__ u32 full_mismatches(const u32 *__restrict__ tplanes, i64 nwords, u32 o,
                                               const uint4 *__restrict__ pq, int NW, bool use_n, u32 tailmask,
                                               u32 budget) {
    const u32 wi = o >> 5, sh = o & 31;
    const u32 *p0 = tplanes + wi, *p1 = tplanes + nwords + wi, *p2 = tplanes + 2 * nwords + wi;
    u32 cnt = 0;
    for (int j = 0; j < NW; ++j) {
        const uint4 q = pq[j];
        u32 x = (__builtin_amdgcn_alignbit(p0[j + 1], p0[j], sh) ^ q.x) |
                (__builtin_amdgcn_alignbit(p1[j + 1], p1[j], sh) ^ q.y);
        if (use_n) x |= (__builtin_amdgcn_alignbit(p2[j + 1], p2[j], sh) ^ q.z);
        if (j == NW - 1) x &= tailmask;
        cnt += __popc(x);
        if (cnt > budget) break;
    }
    return cnt;
}

// ------------------------------------------------------------------------
// K1c: seed filter for the same conditions as the tiled scan (pigeonhole
// anchors, lcf_thres == L, DNA alphabet).  Under those conditions a covered
// (probe, offset) has at least one anchor k-mer that matches exactly
// (SURVEY.md App. A.8), which is how the reference itself finds it
// (catch/probe.py:1062-1069).  Instead of testing every probe at every offset
// (O(P*G)) each target position looks its k-mer up in a hash table of the
// anchor k-mers (O(G)) and only the seeded (probe, offset) pairs are verified
// with the exact plane-wise Hamming test.  Keys are the first min(k,32) bases
// of the k-mer on planes 0/1 (N reads as A): a superset filter, the
// verification is exact.  A covered pair is usually seeded by several
// anchors; it is emitted only from the lowest exact anchor.
// ------------------------------------------------------------------------
#include "rows_bucket.inc"

// A slot is 16 bytes: {key (8 B; open addressing, EMPTY = ~0), first index into
// ents, count} -- the look-up's probe per target position is ONE 16-byte gather
// (key and range used to be two arrays: two dependent gathers per position).
struct SeedTable {
    uint4 *slot;                // {key lo, key hi, first, count}
    u32 *cnt;                   // per slot: entries with this key (build time)
    u32 *ents;                  // entry ids (probe * nanchor + anchor) grouped by key
    u32 mask;                   // capacity - 1
    // anchor-pair filter (seed_lookup_kernel; null when it does not apply): per entry (p, a) ONE 16-byte record
    // {entry id, table slot of the probe's other anchors in anchor order (SEED_SIB of them; bit 31 = the anchor holds
    // an N)}.  Round 3: the slots, not the keys -- every anchor's k-mer has a slot (those that stay out of the
    // table as a key without entries), a key has exactly one slot, so "the target's k-mer there equals that
    // anchor" is a comparison of two slot numbers, and the record is half as long (28 of the look-up's bytes per
    // table match were these records).
    const uint4 *sib;
    // presence bits (round 3; null = off): 4 bits per slot, one of them set per key of the table (a second hash).
    // 24 of 25 target positions hold no anchor k-mer, and each such miss used to cost a 64-byte sector of the
    // 16-byte slots (half a gigabyte of them: HBM); the bits of the same table are 16 MB and stay in the
    // memory-side cache.
    u32 *present;
    u32 pmask;                  // presence bits - 1
    __device__ __forceinline__ unsigned long long *key_at(u32 s) const { return (unsigned long long *)(slot + s); }
};
#define SEED_EMPTY 0xffffffffffffffffull
#define SEED_DEAD 0xffffffffu   // work-list entry without a seed (the tail of a look-up workgroup's range)
#define SEED_SIB 3              // other anchors per entry: the filter takes tables of <= 4 anchors per probe
#define SEED_RSIB 2             // random anchor tables: the anchors just below the entry's that the filter looks at
#define SEED_KEYBITS 0x3fffffffffffffffull
#define SEED_SLOT_N 0x80000000u      // sibling / target slot word: the k-mer holds an N
#define SEED_SLOT_BITS 0x7fffffffu
#define SEED_SLOT_NONE 0x7fffffffu   // target: no such key in the table / no k-mer here
#define SEED_SLOT_ABSENT 0x7ffffffeu // sibling: the probe has no such anchor

__device__ __forceinline__ u32 seed_hash(unsigned long long k) {
    k ^= k >> 29; k *= 0xbf58476d1ce4e5b9ull; k ^= k >> 32;
    return (u32)k;
}
__device__ __forceinline__ u32 seed_hash2(unsigned long long k) {
    k ^= k >> 31; k *= 0x94d049bb133111ebull; k ^= k >> 29;
    return (u32)k;
}

// exclusive prefix sum over the 64 lanes of a wave; *total = wave sum
__device__ __forceinline__ u32 wave_excl_scan(u32 v, u32 *total) {
    const int lane = __lane_id();
    u32 incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    *total = __shfl(incl, 63);
    return incl - v;
}

// 2-plane key of kb bases starting at base offset `o` of a plane pair
__device__ __forceinline__ unsigned long long plane_key(const u32 *__restrict__ p0, const u32 *__restrict__ p1,
                                                        u32 o, int kb) {
    const u32 wi = o >> 5, sh = o & 31;
    const u32 m = kb >= 32 ? 0xffffffffu : ((1u << kb) - 1u);
    const u32 a = __builtin_amdgcn_alignbit(p0[wi + 1], p0[wi], sh) & m;
    const u32 b = __builtin_amdgcn_alignbit(p1[wi + 1], p1[wi], sh) & m;
    return ((unsigned long long)b << 32) | a;
}

// the same with bit 63 set when plane 2 (N) is set anywhere in the k-mer
__device__ __forceinline__ unsigned long long plane_key_n(const u32 *__restrict__ tplanes, i64 nwords, u32 o, int kb,
                                                          int has_n) {
    unsigned long long key = plane_key(tplanes, tplanes + nwords, o, kb);
    if (has_n) {
        const u32 wi = o >> 5, sh = o & 31;
        const u32 m = kb >= 32 ? 0xffffffffu : ((1u << kb) - 1u);
        const u32 *p2 = tplanes + 2 * nwords;
        if (__builtin_amdgcn_alignbit(p2[wi + 1], p2[wi], sh) & m) key |= 1ull << 63;
    }
    return key;
}

// table build 1/3: claim the key's slot, count the entries per slot.  Entries
// are the probes' anchors sorted by (probe, position); with pigeonhole anchors
// only positions below pos_limit enter the table (see run_seed_async).
#define SEED_SKIP 0xffffffffu
__global__ void __launch_bounds__(256)
seed_count_kernel(const uint4 *__restrict__ pplanes, const u32 *__restrict__ ent_probe,
                  const u32 *__restrict__ ent_pos, u32 nent, u32 pos_limit, int nanch, int k, int NW, int kb,
                  SeedTable t, u32 *__restrict__ slot_of, u32 *__restrict__ aslot) {
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nent) return;
    const u32 p = nanch ? e / (u32)nanch : ent_probe[e];
    const u32 o = nanch ? (e % (u32)nanch) * (u32)k : ent_pos[e];
    if (o >= pos_limit && !aslot) { slot_of[e] = SEED_SKIP; return; }
    // the probe image is [word][4]: planes 0/1 of the two words the k-mer starts in
    const u32 wi = o >> 5, sh = o & 31;
    const uint4 w0 = pplanes[(size_t)p * NW + wi];
    const uint4 w1 = (int)(wi + 1) < NW ? pplanes[(size_t)p * NW + wi + 1] : make_uint4(0, 0, 0, 0);
    const u32 m = kb >= 32 ? 0xffffffffu : ((1u << kb) - 1u);
    const unsigned long long key = ((unsigned long long)(__builtin_amdgcn_alignbit(w1.y, w0.y, sh) & m) << 32) |
                                   (__builtin_amdgcn_alignbit(w1.x, w0.x, sh) & m);
    u32 s = seed_hash(key) & t.mask;
    bool fresh = false;
    for (;;) {
        // (round 6: a look first -- a k-mer is the anchor of ~120 probes on S4, and all but the first of them used to pay
        // a compare-and-swap on the one address that holds it)
        unsigned long long prev = *(volatile unsigned long long *)t.key_at(s);
        if (prev == SEED_EMPTY) prev = atomicCAS(t.key_at(s), SEED_EMPTY, key);
        else if (prev != key) { s = (s + 1) & t.mask; continue; }
        fresh = prev == SEED_EMPTY;
        if (fresh || prev == key) break;
        s = (s + 1) & t.mask;
    }
    if (t.present && fresh) {                        // (whoever creates the key's slot sets its presence bit)
        const u32 b = seed_hash2(key) & t.pmask;
        atomicOr(&t.present[b >> 5], 1u << (b & 31u));
    }
    // (the anchor-pair filter wants the slot of EVERY anchor's k-mer, also of those that stay out of the table
    // -- a key without entries; bit 31: the anchor holds an N -- it then equals no target k-mer exactly)
    if (aslot) aslot[e] = s | ((__builtin_amdgcn_alignbit(w1.z, w0.z, sh) & m) ? SEED_SLOT_N : 0u);
    if (o >= pos_limit) { slot_of[e] = SEED_SKIP; return; }
    slot_of[e] = s;
    atomicAdd(&t.cnt[s], 1u);
}

// one launch instead of five memsets
__global__ void __launch_bounds__(256)
seed_init_kernel(uint4 *__restrict__ slot, u32 *__restrict__ cnt, u32 capacity, u32 *__restrict__ ctr,
                 u32 *__restrict__ bcnt, u32 nb, u32 *__restrict__ res, u32 *__restrict__ present, u32 npresent_words) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (u32 i = t; i < capacity; i += stride) { slot[i] = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u); cnt[i] = 0; }
    for (u32 i = t; i < npresent_words; i += stride) present[i] = 0u;
    for (u32 i = t; i < nb; i += stride) bcnt[i] = 0;
    if (t < 8) { ctr[t & 3] = 0; res[t] = 0; }
}

// table build 2/3: give every used slot a contiguous range of ents[]
// (one cursor atomic per 1024 slots: same-address atomics cost ~10 ns each).
// 256 threads x 4 slots, not 1024 x 1: with other groups' kernels resident a
// 16-wave workgroup waits for a whole CU's worth of free wave slots, and this
// kernel then took milliseconds (S4, four groups in flight: 3.5 ms on average,
// 33 ms at worst, 16 % of all kernel time).
#define SA_SLOTS 1024
__global__ void __launch_bounds__(256)
seed_alloc_kernel(SeedTable t, u32 *__restrict__ cursor) {
    __shared__ u32 s_part[4], s_base;
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 s0 = blockIdx.x * SA_SLOTS + threadIdx.x * 4;
    uint4 n = make_uint4(0, 0, 0, 0);
    if (s0 + 3 <= t.mask) n = *(const uint4 *)(t.cnt + s0);   // the table size is a multiple of 1024
    const u32 mine = n.x + n.y + n.z + n.w;
    u32 total;
    const u32 ex = wave_excl_scan(mine, &total);
    if (lane == 0) s_part[wave] = total;
    __syncthreads();
    u32 woff = 0, tot = 0;
    for (int w = 0; w < 4; ++w) { if (w < (int)wave) woff += s_part[w]; tot += s_part[w]; }
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(cursor, tot) : 0u;
    __syncthreads();
    if (s0 + 3 <= t.mask) {
        const u32 b = s_base + woff + ex;
        uint2 *r = (uint2 *)(t.slot + s0);       // the (first, count) half of slots s0 .. s0 + 3
        r[1] = make_uint2(b, n.x);
        r[3] = make_uint2(b + n.x, n.y);
        r[5] = make_uint2(b + n.x + n.y, n.z);
        r[7] = make_uint2(b + n.x + n.y + n.z, n.w);
    }
}

// table build 3/3: drop the entries into their slot's range
__global__ void __launch_bounds__(256)
seed_fill_kernel(u32 nent, SeedTable t, const u32 *__restrict__ slot_of, int nanch,
                 const u32 *__restrict__ aslot, uint4 *__restrict__ sib,
                 const u32 *__restrict__ ent_probe, const u32 *__restrict__ ent_pos) {
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nent) return;
    const u32 s = slot_of[e];
    if (s == SEED_SKIP) return;
    const u32 j = atomicSub(&t.cnt[s], 1u) - 1u;
    const u32 idx = t.slot[s].z + j;
    t.ents[idx] = e;
    if (sib && !nanch) {
        // random anchors (entries sorted by (probe, position)): the slots of the probe's two anchors just below
        // this one and how far below they start (<= L - k <= SL_HALO < 256)
        const u32 p = ent_probe[e], o = ent_pos[e];
        u32 kk[SEED_RSIB], dd = 0;
        for (u32 q = 0; q < SEED_RSIB; ++q) {
            kk[q] = SEED_SLOT_ABSENT;
            if (e > q && ent_probe[e - 1 - q] == p) { kk[q] = aslot[e - 1 - q]; dd |= (o - ent_pos[e - 1 - q]) << (8 * q); }
        }
        sib[idx] = make_uint4(e, kk[0], kk[1], dd);
    } else if (sib) {
        const u32 a = e % (u32)nanch, e0 = e - a;
        u32 kk[SEED_SIB];
        u32 q = 0;
        for (u32 b = 0; b < (u32)nanch; ++b)
            if (b != a && q < SEED_SIB) kk[q++] = aslot[e0 + b];
        for (; q < SEED_SIB; ++q) kk[q] = SEED_SLOT_ABSENT;
        sib[idx] = make_uint4(e, kk[0], kk[1], kk[2]);      // (a = e % nanch)
    }
}

// scan 1/2: every target position looks its k-mer up; the matching anchors
// are expanded into the seed work list (position, entry, sequence).  A
// workgroup owns SL_TILE positions: one cursor atomic per workgroup, and the
// seeds (a few positions hold dozens, most none) are written by all threads
// through a prefix-sum + binary search in LDS.
#define SL_THREADS 512
#define SL_PPT 4
#define SL_TILE (SL_THREADS * SL_PPT)
#define SL_HALO 96   // target keys kept on either side of the tile: (anchors per probe - 1) * k <= 90
// The anchor-pair filter (t.sib; pigeonhole tables with A = L / k <= 4 anchors per probe, k <= 30, A - m >= 2).
// A window with <= m mismatches leaves at least A - m of the A disjoint anchors exact, and it is reported
// from its LOWEST exact anchor only.  So the seed of anchor a at position i (window start i - a k) is needed
// only if (1) no lower anchor b < a of the probe is exact -- the target's k-mer at i - (a - b) k differs from
// that anchor's key or one of them holds an N (key flags, bits 62 / 63): otherwise the pair is reported from
// b, whose own seed passes this test by induction down to the lowest exact anchor -- and (2) with A - m >= 2,
// some HIGHER anchor matches on planes 0/1 as well (a superset of "is exact"): a window whose only matching
// anchor is a has more than m mismatches.  The keys of a probe's other anchors sit beside the table entry,
// the target's keys of the tile (+ SL_HALO positions either side) in LDS.  On S4 (-m 2, A = 4) this keeps
// 41 % of the seeds; every true pair keeps exactly one.  The survivors are written densely at the front of
// the workgroup's reservation in the list (sizes are fixed before the entries are read) and ranges[] tells
// the verify kernel where they are: a workgroup of it walks one range, so nothing is launched for the rest.
__global__ void __launch_bounds__(SL_THREADS)
seed_lookup_kernel(const u32 *__restrict__ tplanes, i64 nwords, u32 total, const u32 *__restrict__ seq_off, u32 nseq,
                   int k, int kb, int nanch, int need2, int t_has_n, SeedTable t, u32 *__restrict__ seed_pos,
                   u32 *__restrict__ seed_ent, u32 *__restrict__ seed_seq, u32 *__restrict__ seed_count, u32 seed_cap,
                   uint2 *__restrict__ ranges) {
    __shared__ u32 s_off[SL_TILE + 1], s_rx[SL_TILE], s_sq[SL_TILE], s_part[SL_THREADS / 64], s_base;
    __shared__ u32 s_slot[SL_TILE + 2 * SL_HALO];   // the table slot of every position's k-mer (filtered look-up)
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 tile0 = blockIdx.x * SL_TILE;
    // the tile's first sequence, found once (the same addresses for every lane); a
    // seeded position then walks on from there -- a tile rarely spans more than one
    // or two sequence ends -- instead of a binary search of its own
    const u32 sq0 = find_segment(seq_off, nseq, min(tile0, total - 1));
    // the slot of a k-mer (SEED_SLOT_NONE: not a key of the table), bit 31: it holds an N
    auto slot_word = [&](u32 pos, u32 found) {
        u32 w = found;
        if (t_has_n && (plane_key_n(tplanes, nwords, pos, kb, 1) >> 63)) w |= SEED_SLOT_N;
        return w;
    };
    auto find_slot = [&](unsigned long long key, uint2 *range) {
        if (t.present) {
            const u32 b = seed_hash2(key) & t.pmask;
            if (!((t.present[b >> 5] >> (b & 31u)) & 1u)) return SEED_SLOT_NONE;
        }
        u32 sl_i = seed_hash(key) & t.mask;
        for (;;) {
            const uint4 sl = t.slot[sl_i];   // written by the table-build launches
            const unsigned long long ks = ((unsigned long long)sl.y << 32) | sl.x;
            if (ks == key) { if (range) *range = make_uint2(sl.z, sl.w); return sl_i; }
            if (ks == SEED_EMPTY) return SEED_SLOT_NONE;
            sl_i = (sl_i + 1) & t.mask;
        }
    };
    if (t.sib && tid < 2 * SL_HALO) {                // the positions either side of the tile
        const long long pos = tid < SL_HALO ? (long long)tile0 - SL_HALO + tid : (long long)tile0 + SL_TILE + (tid - SL_HALO);
        const bool ok = pos >= 0 && pos + k <= (long long)total;
        u32 w = SEED_SLOT_NONE;
        if (ok) {
            const u32 f = find_slot(plane_key(tplanes, tplanes + nwords, (u32)pos, kb), nullptr);
            w = f == SEED_SLOT_NONE ? f : slot_word((u32)pos, f);
        }
        s_slot[tid < SL_HALO ? tid : SL_TILE + tid] = w;
    }
    u32 cnt[SL_PPT], mine = 0;
#pragma unroll
    for (int j = 0; j < SL_PPT; ++j) {
        const u32 q = tid * SL_PPT + j, i = tile0 + q;
        uint2 r = make_uint2(0, 0);
        if (t.sib) s_slot[SL_HALO + q] = SEED_SLOT_NONE;
        if (i < total && i + (u32)k <= total) {
            const unsigned long long key = plane_key(tplanes, tplanes + nwords, i, kb);
            const u32 f = find_slot(key, &r);
            if (t.sib && f != SEED_SLOT_NONE) s_slot[SL_HALO + q] = slot_word(i, f);
        }
        if (r.y) {
            u32 sq = sq0;
            while (sq + 1 < nseq && seq_off[sq + 1] <= i) ++sq;
            s_sq[q] = sq;
            if (i + (u32)k > seq_off[sq + 1]) r.y = 0;   // the k-mer must lie inside one sequence
        }
        s_rx[q] = r.x;
        cnt[j] = r.y;
        mine += r.y;
    }
    u32 wtotal;
    const u32 ex = wave_excl_scan(mine, &wtotal);
    if (lane == 0) s_part[wave] = wtotal;
    __syncthreads();
    u32 woff = 0, tot = 0;
    for (int w = 0; w < SL_THREADS / 64; ++w) { if (w < (int)wave) woff += s_part[w]; tot += s_part[w]; }
    if (tot == 0) { if (t.sib && tid == 0) ranges[blockIdx.x] = make_uint2(0u, 0u); return; }
    u32 run = woff + ex;
#pragma unroll
    for (int j = 0; j < SL_PPT; ++j) { s_off[tid * SL_PPT + j] = run; run += cnt[j]; }
    // (filtered: ranges of whole 64-entry groups, as the verify kernel files its hits per group of 64)
    if (tid == 0) { s_off[SL_TILE] = tot; s_base = atomicAdd(seed_count, t.sib ? (tot + 63u) & ~63u : tot); }
    __syncthreads();
    const u32 base = s_base;
    if (!t.sib) {
        if (tid == 0) atomicAdd(seed_count + 1, tot);   // ctr[2]: seeds to verify
        for (u32 d = tid; d < tot; d += SL_THREADS) {
            // last position q with s_off[q] <= d (positions without seeds share the next one's offset)
            u32 lo = 0, hi = SL_TILE;
            while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_off[mid] <= d) lo = mid; else hi = mid; }
            const u32 o = base + d;
            if (o < seed_cap) {
                seed_pos[o] = tile0 + lo;
                seed_ent[o] = t.ents[s_rx[lo] + (d - s_off[lo])];
                seed_seq[o] = s_sq[lo];
            }
        }
        return;
    }
    // filtered: 512 candidates at a time, the kept ones appended in order
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    u32 out_run = 0;
    for (u32 c0 = 0; c0 < tot; c0 += SL_THREADS) {
        const u32 d = c0 + tid;
        bool keep = false;
        u32 lo = 0, ent = 0;
        if (d < tot) {
            u32 hi = SL_TILE;
            while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_off[mid] <= d) lo = mid; else hi = mid; }
            const u32 idx = s_rx[lo] + (d - s_off[lo]);
            const uint4 r0 = t.sib[idx];
            ent = r0.x;
            bool lower_exact = false, higher = need2 == 0;
            if (!nanch) {
                // random anchors: is one of the probe's SEED_RSIB anchors just below this one exact in the same window?
                const u32 rk[SEED_RSIB] = {r0.y, r0.z};
#pragma unroll
                for (u32 j = 0; j < SEED_RSIB; ++j) {
                    const u32 pk = rk[j];
                    const int off = (int)lo - (int)((r0.w >> (8 * j)) & 0xffu) + SL_HALO;    // (>= 0: the distance is <= SL_HALO)
                    const u32 tk = s_slot[off];
                    const bool same = pk != SEED_SLOT_ABSENT && (tk & SEED_SLOT_BITS) != SEED_SLOT_NONE && ((tk ^ pk) & SEED_SLOT_BITS) == 0u;
                    lower_exact = lower_exact || (same && !((tk | pk) & SEED_SLOT_N));
                }
            }
            const u32 a = nanch ? ent % (u32)nanch : 0u;  // the entry's anchor index
            const u32 sk[SEED_SIB] = {r0.y, r0.z, r0.w};
#pragma unroll
            for (u32 j = 0; j < SEED_SIB; ++j) {
                if (!nanch) break;
                const u32 b = j < a ? j : j + 1;
                if (b >= (u32)nanch) continue;
                const int off = (int)lo + ((int)b - (int)a) * k + SL_HALO;
                const u32 tk = (off >= 0 && off < SL_TILE + 2 * SL_HALO) ? s_slot[off] : SEED_SLOT_NONE;
                const u32 pk = sk[j];
                // same slot = same k-mer on planes 0/1; exactly equal only if neither holds an N
                const bool same = (tk & SEED_SLOT_BITS) != SEED_SLOT_NONE && ((tk ^ pk) & SEED_SLOT_BITS) == 0u;
                if (b < a) lower_exact = lower_exact || (same && !((tk | pk) & SEED_SLOT_N));
                else higher = higher || same;
            }
            keep = !lower_exact && higher;
        }
        const unsigned long long bal = __ballot(keep);
        __syncthreads();                 // (the previous chunk is done with s_part)
        if (lane == 0) s_part[wave] = (u32)__popcll(bal);
        __syncthreads();
        u32 wo = 0, ctot = 0;
        for (int w = 0; w < SL_THREADS / 64; ++w) { if (w < (int)wave) wo += s_part[w]; ctot += s_part[w]; }
        if (keep) {
            const u32 o = base + out_run + wo + (u32)__popcll(bal & lt);
            if (o < seed_cap) { seed_pos[o] = tile0 + lo; seed_ent[o] = ent; seed_seq[o] = s_sq[lo]; }
        }
        out_run += ctot;
    }
    // the verify kernel walks ranges[]: [base, base + kept) of every look-up workgroup; the rest of the
    // reservation is never touched
    if (tid == 0) {
        ranges[blockIdx.x] = make_uint2(base, out_run);
        if (out_run) atomicAdd(seed_count + 1, out_run);   // ctr[2]: seeds to verify
    }
}

// bits [pos, pos+len) of the NW-word mismatch mask are all zero
template <int NW>
__device__ __forceinline__ bool mask_range_zero(const u32 (&mw)[NW], int pos, int len) {
    u32 acc = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int lo = max(pos - 32 * j, 0), hi = min(pos + len - 32 * j, 32);
        if (hi > lo) {
            const u32 m = (hi - lo >= 32 ? 0xffffffffu : ((1u << (hi - lo)) - 1u)) << lo;
            acc |= mw[j] & m;
        }
    }
    return acc == 0;
}

// scan 2/2: one thread per seed verifies the (probe, offset) pair exactly and
// files the hit with its bucket (rows_bucket.inc)
template <int NW>
__global__ void __launch_bounds__(256)
seed_verify_kernel(const u32 *__restrict__ tplanes, i64 nwords, const u32 *__restrict__ seq_off,
                   const uint4 *__restrict__ pplanes, const u32 *__restrict__ ent_probe,
                   const u32 *__restrict__ ent_pos, const u32 *__restrict__ ent_ptr, int nanch, int L, int k, int mm,
                   u32 tailmask, int use_n, const u32 *__restrict__ seed_pos, const u32 *__restrict__ seed_ent,
                   const u32 *__restrict__ seed_seq, const u32 *__restrict__ seed_count, u32 seed_cap,
                   HitSink sink) {
    const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= min(*seed_count, seed_cap)) return;
    const u32 e0 = seed_ent[d];
    const bool alive = e0 != SEED_DEAD;   // (dead: the unused tail of a look-up workgroup's range)
    const u32 i = alive ? seed_pos[d] : 0u, e = alive ? e0 : 0u, sq = alive ? seed_seq[d] : 0u;
    // pigeonhole tables (nanch = L/k anchors per probe, sorted) need no look-ups
    const u32 p = nanch ? e / (u32)nanch : ent_probe[e];
    const u32 apos = nanch ? (e % (u32)nanch) * (u32)k : ent_pos[e];
    const u32 lo = seq_off[sq], hi = seq_off[sq + 1];
    bool ok = alive && i >= lo + apos;
    if (sink.probe_group) ok = ok && sink.probe_group[p] == sink.seq_group[sq];   // another instance's sequence
    const u32 o = i - apos;                     // where the probe would start
    ok = ok && o + (u32)L <= hi;                // window inside this sequence
    if (ok) {
        const u32 wi = o >> 5, sh = o & 31;
        const u32 *p0 = tplanes + wi, *p1 = tplanes + nwords + wi, *p2 = tplanes + 2 * nwords + wi;
        const uint4 *pq = pplanes + (size_t)p * NW;
        u32 mw[NW];
        u32 cnt = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const uint4 q = pq[j];
            u32 x = (__builtin_amdgcn_alignbit(p0[j + 1], p0[j], sh) ^ q.x) |
                    (__builtin_amdgcn_alignbit(p1[j + 1], p1[j], sh) ^ q.y);
            if (use_n) x |= (__builtin_amdgcn_alignbit(p2[j + 1], p2[j], sh) ^ q.z);
            if (j == NW - 1) x &= tailmask;
            mw[j] = x;
            cnt += __popc(x);
        }
        ok = cnt <= (u32)mm;
        // the seeding anchor must be exact on all planes (the key ignores plane 2 and
        // bases beyond 32), and the pair is reported from its lowest exact anchor only
        // (the probe's anchors are sorted by position; lower ones precede entry e)
        ok = ok && mask_range_zero<NW>(mw, (int)apos, k);
        if (nanch) {
            for (u32 b = 0; ok && b * (u32)k < apos; ++b)
                if (mask_range_zero<NW>(mw, (int)(b * k), k)) ok = false;
        } else {
            for (u32 j = ent_ptr[p]; ok && j < e; ++j)
                if (mask_range_zero<NW>(mw, (int)ent_pos[j], k)) ok = false;
        }
    }
    hit_file(sink, d, ok, p, o, o + (u32)L, sq, lo, hi);
}

// ---- cooperative verify: 4 lanes per seed ---------------------------------
// seed_verify_kernel above gathers ~30 scattered dwords per lane (3 planes x
// (NW+1) target words + the probe image): every wave-level load touches 64
// different cache lines and the kernel is bound by the vector L1's line rate
// (S4: 1.8e9 seeds x ~30 line look-ups / (256 CUs x 1 line per clock) = 87 ms,
// measured 99).  Here a group of 4 lanes owns one seed: lane t loads words 2t
// and 2t+1 of the window from the word-interleaved target image (two adjacent
// 16-byte words: planes 0,1,2 of 32 bases each) and of the probe image, so a
// wave-level load covers 16 seeds x one contiguous run of <= 128 bytes.  The
// funnel shift takes the following word from the neighbouring lane (DPP
// row_shl:1), the mismatch count is a 2-step DPP butterfly inside the quad,
// the anchor tests are ballots against per-lane constant masks (pigeonhole
// anchors sit at multiples of k whatever the seed).  A wavefront takes 64
// consecutive seeds: it loads their work items coalesced, verifies them 16 at
// a time and files the hits with all 64 lanes.  NW <= 7 (the window needs
// NW + 1 <= 8 words).  (A first version with 8 lanes per seed and the masks
// computed per seed was issue-bound: 99 -> 65 ms, ~24 instructions per seed.)
__device__ __forceinline__ u32 dpp_row_shl1(u32 v) {   // lane i <- lane i + 1 (inside a row of 16)
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);
}
__device__ __forceinline__ u32 quad_sum(u32 v) {       // sum over the 4 lanes of a quad, in every lane
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    return v;
}
// bits [pos, pos+len) of a (NW x 32)-bit string restricted to word t
__device__ __forceinline__ u32 word_range_mask(int t, int pos, int len) {
    const int lo = max(pos - 32 * t, 0), hi = min(pos + len - 32 * t, 32);
    if (hi <= lo) return 0u;
    return (hi - lo >= 32 ? 0xffffffffu : ((1u << (hi - lo)) - 1u)) << lo;
}
#define SV_AMAX 4   // pigeonhole anchors with precomputed masks (more: masks on the fly)

template <int NW>
__global__ void __launch_bounds__(256)
seed_verify4_kernel(const uint4 *__restrict__ tq, const u32 *__restrict__ seq_off,
                    const uint4 *__restrict__ pplanes, const u32 *__restrict__ ent_probe,
                    const u32 *__restrict__ ent_pos, const u32 *__restrict__ ent_ptr, int nanch, int ntab, int L, int k,
                    int mm, u32 tailmask, int use_n, const u32 *__restrict__ seed_pos,
                    const u32 *__restrict__ seed_ent, const u32 *__restrict__ seed_seq,
                    const u32 *__restrict__ seed_count, u32 seed_cap, HitSink sink, const uint2 *__restrict__ ranges) {
    static_assert(NW >= 1 && NW <= 7, "the window needs NW + 1 <= 8 words");
    const u32 lane = threadIdx.x & 63, sub = lane & 3, grp = lane >> 2;
    // ranges (filtered look-up): this workgroup walks the seeds [base, base + kept) of look-up workgroup
    // blockIdx.x, 256 at a time; otherwise every wavefront takes the 64 list entries of its global index
    u32 d0, nseeds = min(*seed_count, seed_cap);
    if (ranges) {
        const uint2 r = ranges[blockIdx.x];
        d0 = r.x + (threadIdx.x >> 6) * 64u;
        nseeds = min(nseeds, r.x + r.y);
    } else {
        d0 = (u32)((((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) << 6);
    }
    // per-lane constants: which of the window's words this lane owns, and the
    // pigeonhole anchors' bit ranges inside them
    const u32 wa = 2 * sub, wb = 2 * sub + 1;
    const bool va = wa < (u32)NW, vb = wb < (u32)NW;
    u32 ma[SV_AMAX], mb[SV_AMAX];
#pragma unroll
    for (int a = 0; a < SV_AMAX; ++a) {
        ma[a] = (nanch && a < ntab) ? word_range_mask((int)wa, a * k, k) : 0u;
        mb[a] = (nanch && a < ntab) ? word_range_mask((int)wb, a * k, k) : 0u;
    }
    // (ranges: the next 64 work items of this wavefront are requested while the current ones are verified --
    // a pass of the loop is a chain of dependent round trips, and this takes one off it)
    u32 n_i = 0, n_e = SEED_DEAD, n_sq = 0;
    if (d0 + lane < nseeds) { n_i = seed_pos[d0 + lane]; n_e = seed_ent[d0 + lane]; n_sq = seed_seq[d0 + lane]; }
  for (; d0 < nseeds; d0 += 256u) {
    // ---- A: this lane's own work item (coalesced) -------------------------
    const u32 d = d0 + lane;
    u32 i = 0, e = 0, sq = 0, p = 0, aidx = 0, apos = 0, lo = 0, hi = 0, o = 0, first_ent = 0;
    bool pre = false;
    const u32 c_i = n_i, c_e = n_e, c_sq = n_sq;
    n_e = SEED_DEAD;
    if (ranges && d + 256u < nseeds) { n_i = seed_pos[d + 256u]; n_e = seed_ent[d + 256u]; n_sq = seed_seq[d + 256u]; }
    if (d < nseeds && c_e != SEED_DEAD) {   // (dead: the unused tail of a look-up workgroup's range)
        i = c_i; e = c_e; sq = c_sq;
        p = nanch ? e / (u32)nanch : ent_probe[e];
        aidx = nanch ? e - p * (u32)nanch : 0u;          // index of the seeding anchor (pigeonhole)
        apos = nanch ? aidx * (u32)k : ent_pos[e];
        lo = seq_off[sq]; hi = seq_off[sq + 1];
        pre = i >= lo + apos;
        if (sink.probe_group) pre = pre && sink.probe_group[p] == sink.seq_group[sq];
        o = i - apos;
        pre = pre && o + (u32)L <= hi;
        if (!nanch && pre) first_ent = ent_ptr[p];
    }
    const unsigned long long premask = __ballot(pre);
    unsigned long long verdict[4] = {0, 0, 0, 0};
    // ---- B: 16 seeds at a time, 4 lanes each -----------------------------
#pragma unroll
    for (u32 it = 0; it < 4; ++it) {
        if (((premask >> (it * 16)) & 0xffffull) == 0) continue;   // wave-uniform
        const u32 src = it * 16 + grp;
        const bool live = (premask >> src) & 1ull;                 // quad-uniform
        const u32 go = __shfl(o, src), gp = __shfl(p, src), ga = __shfl(nanch ? aidx : apos, src);
        u32 x0 = 0, x1 = 0;
        if (live) {
            const u32 wi = go >> 5, sh = go & 31;
            const uint4 *tp = tq + (size_t)wi + wa;                // words wi .. wi+7 (the image has slack words)
            const uint4 T0 = tp[0], T1 = tp[1];
            uint4 Q0 = make_uint4(0, 0, 0, 0), Q1 = make_uint4(0, 0, 0, 0);
            if (va) Q0 = pplanes[(size_t)gp * NW + wa];
            if (vb) Q1 = pplanes[(size_t)gp * NW + wb];
            const u32 nx = dpp_row_shl1(T0.x), ny = dpp_row_shl1(T0.y), nz = dpp_row_shl1(T0.z);
            x0 = (__builtin_amdgcn_alignbit(T1.x, T0.x, sh) ^ Q0.x) | (__builtin_amdgcn_alignbit(T1.y, T0.y, sh) ^ Q0.y);
            x1 = (__builtin_amdgcn_alignbit(nx, T1.x, sh) ^ Q1.x) | (__builtin_amdgcn_alignbit(ny, T1.y, sh) ^ Q1.y);
            if (use_n) {
                x0 |= __builtin_amdgcn_alignbit(T1.z, T0.z, sh) ^ Q0.z;
                x1 |= __builtin_amdgcn_alignbit(nz, T1.z, sh) ^ Q1.z;
            }
            if (wa == (u32)NW - 1) x0 &= tailmask;
            if (wb == (u32)NW - 1) x1 &= tailmask;
            if (!va) x0 = 0;
            if (!vb) x1 = 0;
        }
        const u32 cnt = quad_sum(__popc(x0) + __popc(x1));
        bool ok = live && cnt <= (u32)mm;
        // The seeding anchor must be exact on all planes (the key ignores plane 2
        // and bases beyond 32), and the pair is reported from its lowest exact
        // anchor only (lower anchors precede entry e).
        if (nanch) {
            // zbits: bit a = "anchor a has a mismatch in this window"
            u32 zbits = 0;
#pragma unroll
            for (int a = 0; a < SV_AMAX; ++a) {
                if (a < ntab) {
                    const unsigned long long b = __ballot(((x0 & ma[a]) | (x1 & mb[a])) != 0);
                    zbits |= ((b >> (grp * 4)) & 0xfull) ? (1u << a) : 0u;
                }
            }
            for (int a = SV_AMAX; a < ntab; ++a) {   // tables with more anchors: masks on the fly
                const unsigned long long b = __ballot(((x0 & word_range_mask((int)wa, a * k, k)) |
                                                       (x1 & word_range_mask((int)wb, a * k, k))) != 0);
                zbits |= ((b >> (grp * 4)) & 0xfull) ? (1u << a) : 0u;
            }
            // anchors below ga all broken, anchor ga exact
            ok = ok && (zbits & ((2u << ga) - 1u)) == ((1u << ga) - 1u);
        } else {
            {
                const unsigned long long b = __ballot(((x0 & word_range_mask((int)wa, (int)ga, k)) |
                                                       (x1 & word_range_mask((int)wb, (int)ga, k))) != 0);
                ok = ok && ((b >> (grp * 4)) & 0xfull) == 0;
            }
            const u32 ge = __shfl(e, src), gf = __shfl(first_ent, src);
            u32 j = gf;
            while (__ballot(ok && j < ge)) {
                const int ap = (ok && j < ge) ? (int)ent_pos[j] : 0;
                const unsigned long long b = __ballot(((x0 & word_range_mask((int)wa, ap, k)) |
                                                       (x1 & word_range_mask((int)wb, ap, k))) != 0);
                if (ok && j < ge && ((b >> (grp * 4)) & 0xfull) == 0) ok = false;
                ++j;
            }
        }
        verdict[it] = __ballot(ok);      // quad-uniform: bit 4*grp stands for seed it*16+grp
    }
    // ---- C: every lane files its own seed ----------------------------------
    if (d < nseeds) {
        const u32 q = lane >> 4;
        const unsigned long long v = q == 0 ? verdict[0] : q == 1 ? verdict[1] : q == 2 ? verdict[2] : verdict[3];
        hit_file(sink, d, (v >> ((lane & 15) * 4)) & 1ull, p, o, o + (u32)L, sq, lo, hi);
    }
    if (!ranges) break;
  }
}

typedef void (*seed_verify_fn)(const u32 *, i64, const u32 *, const uint4 *, const u32 *, const u32 *, const u32 *,
                               int, int, int, int, u32, int, const u32 *, const u32 *, const u32 *, const u32 *, u32,
                               HitSink);
static seed_verify_fn pick_seed_verify(int nw) {
    switch (nw) {
    case 1: return seed_verify_kernel<1>;
    case 2: return seed_verify_kernel<2>;
    case 3: return seed_verify_kernel<3>;
    case 4: return seed_verify_kernel<4>;
    case 5: return seed_verify_kernel<5>;
    case 6: return seed_verify_kernel<6>;
    case 7: return seed_verify_kernel<7>;
    case 8: return seed_verify_kernel<8>;
    }
    return nullptr;
}

typedef void (*seed_verify4_fn)(const uint4 *, const u32 *, const uint4 *, const u32 *, const u32 *, const u32 *, int, int,
                                int, int, int, u32, int, const u32 *, const u32 *, const u32 *, const u32 *, u32,
                                HitSink, const uint2 *);
static seed_verify4_fn pick_seed_verify4(int nw) {
    switch (nw) {
    case 1: return seed_verify4_kernel<1>;
    case 2: return seed_verify4_kernel<2>;
    case 3: return seed_verify4_kernel<3>;
    case 4: return seed_verify4_kernel<4>;
    case 5: return seed_verify4_kernel<5>;
    case 6: return seed_verify4_kernel<6>;
    case 7: return seed_verify4_kernel<7>;
    }
    return nullptr;
}

#include "scan_join.inc"

#ifndef CATCHHIP_NO_TILED_SCAN   // (the tests' cross-check scan; see catchhip_cover_scan)
// The tiled scan kernel (see the comment above full_mismatches).
__global__ void __launch_bounds__(SF2_THREADS)
scan_fast3_kernel(const u32 *__restrict__ tplanes, i64 nwords, u32 total, const u32 *__restrict__ seq_off,
                  u32 nseq, const uint2 *__restrict__ pw0, const uint4 *__restrict__ pplanes, u32 nprobes,
                  u32 probes_per_block, int L, int NW, int mm, u32 tailmask, int use_n,
                  u32 *__restrict__ hit_probe, u32 *__restrict__ hit_pos, u32 *__restrict__ hit_count,
                  u32 hit_cap) {
    const int tid = threadIdx.x;
    const u32 tile0 = blockIdx.x * SF2_TILE;
    const u32 p_begin = blockIdx.y * probes_per_block;
    const u32 p_end = min(nprobes, p_begin + probes_per_block);
    const u32 mask0 = NW == 1 ? tailmask : 0xffffffffu;

    u32 T0[SF2_OPL], T1[SF2_OPL];
#pragma unroll
    for (int w = 0; w < SF2_OPL; ++w) {
        const u32 o = tile0 + w * SF2_THREADS + tid;
        const u32 wi = o >> 5, sh = o & 31;
        const u32 *p0 = tplanes + wi, *p1 = tplanes + nwords + wi;
        T0[w] = __builtin_amdgcn_alignbit(p0[1], p0[0], sh) & mask0;
        T1[w] = __builtin_amdgcn_alignbit(p1[1], p1[0], sh) & mask0;
    }
    for (u32 q = p_begin; q < p_end; ++q) {
        const uint2 qq = pw0[q];   // uniform
        u32 mn = 64;
#pragma unroll
        for (int w = 0; w < SF2_OPL; ++w) {
            const u32 x = __builtin_amdgcn_bitop3_b32(T1[w], T0[w] ^ qq.x, qq.y, 0xde);
            mn = min(mn, (u32)__popc(x));
        }
        if (__ballot(mn <= (u32)mm) == 0ull) continue;
        if (mn > (u32)mm) continue;
        const uint4 *pq = pplanes + (size_t)q * NW;
#pragma unroll 1
        for (int w = 0; w < SF2_OPL; ++w) {
            if ((u32)__popc((T0[w] ^ qq.x) | (T1[w] ^ qq.y)) > (u32)mm) continue;
            const u32 o = tile0 + w * SF2_THREADS + tid;
            if (o >= total || o + (u32)L > total) continue;
            if (full_mismatches(tplanes, nwords, o, pq, NW, use_n != 0, tailmask, (u32)mm) > (u32)mm) continue;
            const u32 s = find_segment(seq_off, nseq, o);
            if (o + (u32)L > seq_off[s + 1]) continue;
            const u32 slot = atomicAdd(hit_count, 1u);
            if (slot < hit_cap) { hit_probe[slot] = q; hit_pos[slot] = o; }
        }
    }
}
#endif

// ------------------------------------------------------------------------
// general path
// ------------------------------------------------------------------------
__device__ __forceinline__ u64 kmer_hash(const u8 *__restrict__ p, int k) {
    u64 h = 0xcbf29ce484222325ull;
    for (int i = 0; i < k; ++i) h = (h ^ (u64)p[i]) * 0x100000001b3ull;
    return h;
}

__global__ void __launch_bounds__(256)
anchor_hash_kernel(const u8 *__restrict__ pbytes, const u32 *__restrict__ probe_off,
                   const i32 *__restrict__ ent_probe, const i32 *__restrict__ ent_pos, u32 nent,
                   int k, u64 *__restrict__ keys, u32 *__restrict__ vals) {
    u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nent) return;
    keys[e] = kmer_hash(pbytes + probe_off[ent_probe[e]] + ent_pos[e], k);
    vals[e] = e;
}

// one lane per target position i: every anchor whose k-mer hash equals the
// hash of seq[i:i+k] becomes a seed hit (entry, i)  (catch/probe.py:1062-1069)
__global__ void __launch_bounds__(256)
seed_join_kernel(const u8 *__restrict__ tbytes, u32 total, const u32 *__restrict__ seq_off,
                 u32 nseq, int k, const u64 *__restrict__ keys, const u32 *__restrict__ vals,
                 u32 nent, HitBuf hb) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total || i + (u32)k > total) return;
    u32 s = find_segment(seq_off, nseq, i);
    if (i + (u32)k > seq_off[s + 1]) return;  // k-mer must lie inside one sequence
    u64 h = kmer_hash(tbytes + i, k);
    u32 lo = 0, hi = nent;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if (keys[mid] < h) lo = mid + 1; else hi = mid;
    }
    for (u32 j = lo; j < nent && keys[j] == h; ++j) {
        u32 slot = atomicAdd(hb.count, 1u);
        if (slot < hb.cap) { hb.a[slot] = vals[j]; hb.b[slot] = i; }
    }
}

#define MAX_MM 32
// one lane per seed hit: the reference's cover function on the aligned window
// (catch/probe.py:1070-1108 + :1328-1344 + longest_common_substring.py:59-159)
__device__ __forceinline__ void extend_bytes(const u8 *__restrict__ tbytes, const u32 *__restrict__ seq_off, u32 s,
                                             const u8 *__restrict__ pbytes, const u32 *__restrict__ probe_off,
                                             i32 p, i64 a, int k, int mm, int lcf_thres, int island, u32 e, u32 gi,
                                             const HitBuf &out) {
    const u8 *pf = pbytes + probe_off[p];
    const i64 L = (i64)(probe_off[p + 1] - probe_off[p]);
    const i64 lo = seq_off[s], G = (i64)seq_off[s + 1] - lo;
    const i64 i = (i64)gi - lo;
    const i64 off = i - a;
    const i64 sub_l = off > 0 ? off : 0;
    const i64 sub_r = (off + L < G) ? off + L : G;
    const i64 W = sub_r - sub_l;              // compared length (both truncated)
    const i64 ps = off < 0 ? -off : 0;        // probe index of window position 0
    const i64 ks = off < 0 ? i : a;           // anchor start in window coordinates
    const i64 ke = ks + k;
    const u8 *x = pf + ps;                    // probe window
    const u8 *y = tbytes + lo + sub_l;        // target window
    // the seed is a k-mer *equality* in the reference (dict lookup): verify it
    for (i64 j = ks; j < ke; ++j)
        if (x[j] != y[j]) return;
    if (mm < 0) return;                       // range(k+1) empty: length -1 < any threshold
    i64 before[MAX_MM + 1], after[MAX_MM + 1];
    int nb = 0, na = 0;
    for (i64 j = ks - 1; j >= 0 && nb <= mm; --j)
        if (x[j] != y[j]) before[nb++] = (ks - 1) - j;
    for (i64 j = ke; j < W && na <= mm; ++j)
        if (x[j] != y[j]) after[na++] = j - ke;
    i64 best_len = -1, best_start = -1;
    for (int q = 0; q <= mm; ++q) {
        i64 bl = (q >= nb) ? ks : before[q];
        i64 al = (mm - q >= na) ? (W - ke) : after[mm - q];
        i64 len = bl + k + al;
        if (len > best_len) { best_len = len; best_start = ks - bl; }
    }
    i64 thr = lcf_thres;
    if (L < thr) thr = L;
    if (G < thr) thr = G;
    if (best_len < thr) return;
    if (island > 0) {
        i64 exact = (nb > 0 ? before[0] : ks) + k + (na > 0 ? after[0] : (W - ke));
        if (mm == 0) exact = best_len;
        if (exact < island) return;
    }
    u32 gs = (u32)(lo + sub_l + best_start);
    u32 slot = atomicAdd(out.count, 1u);
    if (slot < out.cap) {
        out.a[slot] = (u32)p; out.b[slot] = gs; out.c[slot] = gs + (u32)best_len;
        if (out.d) { out.d[slot] = gi; out.e[slot] = e; }
    }
}

__global__ void __launch_bounds__(256)
extend_kernel(const u8 *__restrict__ tbytes, const u32 *__restrict__ seq_off, u32 nseq,
              const u8 *__restrict__ pbytes, const u32 *__restrict__ probe_off,
              const i32 *__restrict__ ent_probe, const i32 *__restrict__ ent_pos, int k,
              int mm, int lcf_thres, int island, const u32 *__restrict__ seed_ent,
              const u32 *__restrict__ seed_pos, u32 nseeds, HitBuf out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseeds) return;
    const u32 e = seed_ent[t], gi = seed_pos[t];
    extend_bytes(tbytes, seq_off, find_segment(seq_off, nseq, gi), pbytes, probe_off, ent_probe[e], (i64)ent_pos[e], k,
                 mm, lcf_thres, island, e, gi, out);
}

// highest set bit of the NW-word mask below position `below`, or -1
template <int NW>
__device__ __forceinline__ int mask_prev(const u32 (&mw)[NW], int below) {
#pragma unroll
    for (int j = NW - 1; j >= 0; --j) {
        const int lim = below - 32 * j;
        if (lim <= 0) continue;
        const u32 w = lim >= 32 ? mw[j] : (mw[j] & ((1u << lim) - 1u));
        if (w) return 32 * j + 31 - __clz((int)w);
    }
    return -1;
}
// lowest set bit at or above position `from`, or -1 (bits beyond the probe are zero)
template <int NW>
__device__ __forceinline__ int mask_next(const u32 (&mw)[NW], int from) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int lo = from - 32 * j;
        if (lo >= 32) continue;
        const u32 w = lo <= 0 ? mw[j] : (mw[j] & ~((1u << lo) - 1u));
        if (w) return 32 * j + __ffs((int)w) - 1;
    }
    return -1;
}

// The same cover function on the packed images (equal-length DNA probes): the
// mismatch mask of the whole window in NW words, then the <= mm+1 mismatches on
// either side of the anchor by bit scans instead of up to L byte comparisons.
// Windows cut by a sequence end (rare) take the byte routine.
struct ExtHit { u32 p, gs, ge, gi, e; };
template <int NW>
__device__ __forceinline__ bool extend_planes_one(u32 t, const u32 *__restrict__ tplanes, i64 nwords,
                                                  const u32 *__restrict__ seq_off, const uint4 *__restrict__ pplanes,
                                                  const i32 *__restrict__ ent_probe, const i32 *__restrict__ ent_pos,
                                                  int L, int k, int mm, int lcf_thres, int island, u32 tailmask,
                                                  int use_n, const u32 *__restrict__ seed_ent,
                                                  const u32 *__restrict__ seed_pos, const u32 *__restrict__ seed_seq,
                                                  u32 *__restrict__ cut, u32 cut_cap, ExtHit &hit) {
    const u32 e = seed_ent[t], gi = seed_pos[t], sq = seed_seq[t];
    const i32 p = ent_probe[e];
    const int a = ent_pos[e];
    const u32 lo = seq_off[sq], hi = seq_off[sq + 1];
    // a window cut by a sequence end goes on a list for the byte routine (its
    // local arrays would give THIS kernel a scratch frame, and with it a fraction of
    // the wavefronts in flight: 15 ms instead of 2 for the seeds of S4 x 0.1)
    const bool is_cut = gi < lo + (u32)a || gi - (u32)a + (u32)L > hi;
    if (is_cut) {
        const u32 at = atomicAdd(&cut[0], 1u);
        if (at < cut_cap) cut[1 + at] = t;
    }
    bool ok = !is_cut && mm >= 0;
    u32 gs = 0, glen = 0;
    if (ok) {
        const u32 o = gi - (u32)a;
        const u32 wi = o >> 5, sh = o & 31;
        const u32 *p0 = tplanes + wi, *p1 = tplanes + nwords + wi, *p2 = tplanes + 2 * nwords + wi;
        const uint4 *pq = pplanes + (size_t)p * NW;
        u32 mw[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const uint4 q = pq[j];
            u32 x = (__builtin_amdgcn_alignbit(p0[j + 1], p0[j], sh) ^ q.x) |
                    (__builtin_amdgcn_alignbit(p1[j + 1], p1[j], sh) ^ q.y);
            if (use_n) x |= (__builtin_amdgcn_alignbit(p2[j + 1], p2[j], sh) ^ q.z);
            if (j == NW - 1) x &= tailmask;
            mw[j] = x;
        }
        // the seed is a k-mer equality in the reference (dict lookup)
        ok = mask_range_zero<NW>(mw, a, k);
        if (ok) {
            const int ks = a, ke = a + k;
            // right side: up to mm+1 mismatches; `rcur` = the na-th of them
            int na = 0, rcur = -1;
            for (int pos = ke; na <= mm;) {
                const int b = mask_next<NW>(mw, pos);
                if (b < 0) break;
                rcur = b; ++na; pos = b + 1;
            }
            // budget q on the left, mm - q on the right (longest_common_substring.py:133-157):
            // the left pointer walks outwards, the right one back in; the first maximum wins
            int best_len = -1, best_start = -1, lcur = ks;
            bool lmore = true;
            for (int q = 0; q <= mm; ++q) {
                int bl = ks;
                if (lmore) {
                    const int b = mask_prev<NW>(mw, lcur);
                    if (b < 0) lmore = false;
                    else { bl = (ks - 1) - b; lcur = b; }
                }
                const int jr = mm - q;                 // index of the right mismatch that stops the run
                int al = L - ke;
                if (jr < na) al = rcur - ke;       // rcur stands on right mismatch number jr
                const int len = bl + k + al;
                if (len > best_len) { best_len = len; best_start = ks - bl; }
                // the next iteration needs right mismatch number jr - 1: the one before rcur
                if (jr < na && jr >= 1) rcur = mask_prev<NW>(mw, rcur);
            }
            i64 thr = lcf_thres;
            if (L < thr) thr = L;
            if ((i64)(hi - lo) < thr) thr = (i64)(hi - lo);
            ok = best_len >= thr;
            if (ok && island > 0) {
                const int b0 = mask_prev<NW>(mw, ks), a0 = mask_next<NW>(mw, ke);
                int exact = (b0 >= 0 ? (ks - 1) - b0 : ks) + k + (a0 >= 0 ? a0 - ke : (L - ke));
                if (mm == 0) exact = best_len;
                ok = exact >= island;
            }
            gs = o + (u32)best_start;
            glen = (u32)best_len;
        }
    }
    hit.p = (u32)p; hit.gs = gs; hit.ge = gs + glen; hit.gi = gi; hit.e = e;
    return ok;
}

// SE_PPT seeds per thread and ONE counter update per workgroup: the hit counter
// is a single address, ~10 ns per atomic -- one per wavefront (what the compiler
// makes of a per-hit atomicAdd) was 1.8 M atomics = 15-20 ms for the 114 M seeds
// of S4 x 0.1, whatever the rest of the kernel did.
#define SE_PPT 4
template <int NW>
__global__ void __launch_bounds__(256)
extend_planes_kernel(const u32 *__restrict__ tplanes, i64 nwords, const u8 *__restrict__ tbytes,
                     const u32 *__restrict__ seq_off, const uint4 *__restrict__ pplanes,
                     const u8 *__restrict__ pbytes, const u32 *__restrict__ probe_off,
                     const i32 *__restrict__ ent_probe, const i32 *__restrict__ ent_pos, int L, int k, int mm,
                     int lcf_thres, int island, u32 tailmask, int use_n, const u32 *__restrict__ seed_ent,
                     const u32 *__restrict__ seed_pos, const u32 *__restrict__ seed_seq, u32 nseeds, HitBuf out,
                     u32 *__restrict__ cut, u32 cut_cap) {
    __shared__ u32 s_part[4], s_base;
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 t0 = blockIdx.x * (256 * SE_PPT);
    ExtHit h[SE_PPT];
    u32 okmask = 0, mine = 0;
#pragma unroll
    for (int q = 0; q < SE_PPT; ++q) {
        const u32 t = t0 + q * 256 + threadIdx.x;   // consecutive lanes, consecutive seeds
        bool ok = false;
        if (t < nseeds)
            ok = extend_planes_one<NW>(t, tplanes, nwords, seq_off, pplanes, ent_probe, ent_pos, L, k, mm, lcf_thres,
                                       island, tailmask, use_n, seed_ent, seed_pos, seed_seq, cut, cut_cap, h[q]);
        okmask |= ok ? (1u << q) : 0u;
        mine += ok ? 1u : 0u;
    }
    u32 wtotal;
    const u32 ex = wave_excl_scan(mine, &wtotal);
    if (lane == 0) s_part[wave] = wtotal;
    __syncthreads();
    u32 woff = 0, tot = 0;
    for (int w = 0; w < 4; ++w) { if (w < (int)wave) woff += s_part[w]; tot += s_part[w]; }
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(out.count, tot) : 0u;
    __syncthreads();
    u32 slot = s_base + woff + ex;
#pragma unroll
    for (int q = 0; q < SE_PPT; ++q) {
        if (!(okmask & (1u << q))) continue;
        if (slot < out.cap) {
            out.a[slot] = h[q].p; out.b[slot] = h[q].gs; out.c[slot] = h[q].ge;
            if (out.d) { out.d[slot] = h[q].gi; out.e[slot] = h[q].e; }
        }
        ++slot;
    }
    (void)tbytes; (void)pbytes; (void)probe_off;
}

// the listed seeds through the byte routine
__global__ void __launch_bounds__(256)
extend_cut_kernel(const u32 *__restrict__ cut, u32 cut_cap, const u8 *__restrict__ tbytes,
                  const u32 *__restrict__ seq_off, const u8 *__restrict__ pbytes, const u32 *__restrict__ probe_off,
                  const i32 *__restrict__ ent_probe, const i32 *__restrict__ ent_pos, int k, int mm, int lcf_thres,
                  int island, const u32 *__restrict__ seed_ent, const u32 *__restrict__ seed_pos,
                  const u32 *__restrict__ seed_seq, HitBuf out) {
    const u32 n = min(cut[0], cut_cap);
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u32 t = cut[1 + i];
        const u32 e = seed_ent[t];
        extend_bytes(tbytes, seq_off, seed_seq[t], pbytes, probe_off, ent_probe[e], (i64)ent_pos[e], k, mm, lcf_thres,
                     island, e, seed_pos[t], out);
    }
}

typedef void (*extend_planes_fn)(const u32 *, i64, const u8 *, const u32 *, const uint4 *, const u8 *, const u32 *,
                                 const i32 *, const i32 *, int, int, int, int, int, u32, int, const u32 *, const u32 *,
                                 const u32 *, u32, HitBuf, u32 *, u32);
static extend_planes_fn pick_extend_planes(int nw) {
    switch (nw) {
    case 1: return extend_planes_kernel<1>;
    case 2: return extend_planes_kernel<2>;
    case 3: return extend_planes_kernel<3>;
    case 4: return extend_planes_kernel<4>;
    case 5: return extend_planes_kernel<5>;
    case 6: return extend_planes_kernel<6>;
    case 7: return extend_planes_kernel<7>;
    case 8: return extend_planes_kernel<8>;
    }
    return nullptr;
}

// ------------------------------------------------------------------------
// rows: extension / clip / sort / merge
// ------------------------------------------------------------------------
// key = (owner set id << 32) | clipped start, val = clipped end
// (catch/filter/set_cover_filter.py:424-439; the genome offset is implicit in
// the global coordinate)
// After the sort: rows of one (set, segment) group are adjacent and ordered by
// start.  The thread standing on the first row of a group walks it, merging
// overlapping and touching intervals (catch/utils/interval.py:288-316);
// `bounds` (genome offsets, or sequence offsets for the tolerant-bp variant)
// define the segments.  head[i] = 1 on the first row of every merged interval,
// whose merged end goes to mend[i].
__global__ void __launch_bounds__(256)
rows_merge_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ vals, u32 n,
                  const u32 *__restrict__ bounds, u32 nb, u32 *__restrict__ head,
                  u32 *__restrict__ mend, u32 *__restrict__ seg) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u64 key = keys[t];
    u32 sid = (u32)(key >> 32), st = (u32)key;
    u32 sg = find_segment(bounds, nb, st);
    seg[t] = sg;
    bool first;
    if (t == 0) first = true;
    else {
        u64 pk = keys[t - 1];
        first = ((u32)(pk >> 32) != sid) || (find_segment(bounds, nb, (u32)pk) != sg);
    }
    if (!first) return;
    u32 hi_bound = bounds[sg + 1];
    u32 cur_head = t, cur_end = vals[t];
    head[t] = 1;
    for (u32 j = t + 1; j < n; ++j) {
        u64 kj = keys[j];
        u32 sj = (u32)kj;
        if ((u32)(kj >> 32) != sid || sj >= hi_bound) break;  // next group
        u32 ej = vals[j];
        if (sj <= cur_end) { head[j] = 0; if (ej > cur_end) cur_end = ej; }
        else { mend[cur_head] = cur_end; cur_head = j; cur_end = ej; head[j] = 1; }
    }
    mend[cur_head] = cur_end;
}

__global__ void __launch_bounds__(256)
rows_compact_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ head,
                    const u32 *__restrict__ mend, const u32 *__restrict__ seg,
                    const u32 *__restrict__ idx, u32 n, i32 *__restrict__ o_set,
                    i32 *__restrict__ o_univ, u32 *__restrict__ o_gs, u32 *__restrict__ o_ge,
                    u32 *__restrict__ lmax) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 len = 0;
    if (t < n && head[t]) {
        u32 d = idx[t];
        o_set[d] = (i32)(keys[t] >> 32);
        o_univ[d] = (i32)seg[t];
        o_gs[d] = (u32)keys[t];
        o_ge[d] = mend[t];
        len = mend[t] - (u32)keys[t];
    }
    // longest row of the table: one atomic per wavefront
    for (int d = 32; d > 0; d >>= 1) { u32 o = __shfl_down(len, d, 64); len = o > len ? o : len; }
    if ((threadIdx.x & 63) == 0 && len) atomicMax(lmax, len);
}

__global__ void __launch_bounds__(256)
rows_bp_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ head,
               const u32 *__restrict__ mend, u32 n, unsigned long long *__restrict__ bp) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || !head[t]) return;
    atomicAdd(&bp[(u32)(keys[t] >> 32)], (unsigned long long)(mend[t] - (u32)keys[t]));
}

// ------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------
struct RawHits {
    DevBuf<u32> a, b, c, count;
    DevBuf<u32> d, e;          // seed position / anchor entry per hit (first-seen scans only)
    bool want_seed = false;
    u32 n = 0;
    bool has_end = false;
};

static int read_count(catchhip_ctx *ctx, const u32 *d, u32 *out) {
    HIP_TRY(hipMemcpyAsync(ctx->h_pin, d, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = *(volatile u32 *)ctx->h_pin;
    return 0;
}

// seed scan: equal-length DNA probes with anchors, full-length cover threshold,
// no island, every sequence at least one probe long (SURVEY.md App. A.8 without
// the pigeonhole requirement: pairs are found through the anchors given)
static bool seed_path_ok(const catchhip_probes *P, const catchhip_targets *T, int mm, int lcf_thres, int island) {
    if (!P->dna5 || !T->dna5) return false;
    if (P->L <= 0 || P->L > 256 || P->pwords < 1 || P->pwords > 8) return false;
    if (P->nent <= 0 || P->k <= 0 || P->k > P->L) return false;
    if (mm < 0) return false;
    if (lcf_thres != P->L || island != 0) return false;
    if (T->nseq == 0 || T->min_seq_len < P->L) return false;
    return true;
}
// tiled scan (every probe at every offset): additionally the anchors must be
// the pigeonhole anchors with L/k > mm, so that some anchor survives any <= mm
// mismatches and the anchor requirement is implied
static bool fast_path_ok(const catchhip_probes *P, const catchhip_targets *T, int mm, int lcf_thres,
                         int island) {
    return seed_path_ok(P, T, mm, lcf_thres, island) && P->pigeonhole && P->L / P->k > mm;
}

static int run_fast(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T, int mm,
                    RawHits &H, PhaseTimer &tm) {
    const bool use_n = P->has_n || T->has_n;
    const u32 tailmask = (P->L & 31) ? ((1u << (P->L & 31)) - 1u) : 0xffffffffu;
    u32 cap = (u32)std::max<i64>((i64)1 << 20, std::min<i64>(P->nprobes * 64, (i64)1 << 28));
    TRY(H.count.alloc(1));
    // enough workgroups to fill 256 CUs several times over
    const u32 ntiles = (u32)div_up(T->total, SF2_TILE);
    u32 want = (u32)div_up((i64)ctx->num_cus * 16, ntiles);
    u32 ppb = (u32)div_up(P->nprobes, want ? want : 1);
    ppb = std::max<u32>(ppb, 64u);
    u32 nchunks = (u32)div_up(P->nprobes, ppb);
    if (nchunks > 65535) { ppb = (u32)div_up(P->nprobes, 65535); nchunks = (u32)div_up(P->nprobes, ppb); }
    for (int attempt = 0; attempt < 3; ++attempt) {
        TRY(H.a.reserve(cap));
        TRY(H.b.reserve(cap));
        HIP_TRY(hipMemsetAsync(H.count.p, 0, sizeof(u32), ctx->stream));
        tm.restart();  // time exactly the scan kernel (HIP events on this stream)
#ifndef CATCHHIP_NO_TILED_SCAN
        hipLaunchKernelGGL(scan_fast3_kernel, dim3(ntiles, nchunks), dim3(SF2_THREADS), 0, ctx->stream,
                           T->planes.p, T->nwords, (u32)T->total, T->seq_off.p, (u32)T->nseq, P->w0.p,
                           (const uint4 *)P->planes.p, (u32)P->nprobes, ppb, (int)P->L, (int)P->pwords, mm,
                           tailmask, use_n ? 1 : 0, H.a.p, H.b.p, H.count.p, cap);
#else
        (void)ntiles; (void)nchunks; (void)ppb; (void)mm; (void)tailmask; (void)use_n;
        chip_set_error("cover_scan: built without the tiled cross-check scan");
        return CATCHHIP_EINVAL;
#endif
        tm.launch();
        tm.stop();
        HIP_TRY(hipGetLastError());
        u32 n;
        TRY(read_count(ctx, H.count.p, &n));
        if (n <= cap) { H.n = n; H.has_end = false; ctx->counters[0] = n; ctx->counters[1] = 0; ctx->seeds_dropped = 0; return 0; }
        cap = n;  // overflow: rerun with the exact size
    }
    chip_set_error("fast scan: hit buffer overflow");
    return CATCHHIP_ENOMEM;
}

// Capacity of the seed work list: 6 seeds per target base covers the data seen
// so far (S4: 3.6 on average, above 4 for some groups); once these probes have
// been scanned the observed ratio (+25 %) sizes the list instead.  An overflow
// is detected on the device and costs a second scan with the exact size.
static u32 seed_capacity(const catchhip_probes *P, const catchhip_targets *T) {
    const double per_base = P->seed_ratio_hint > 0.0 ? std::max(1.0, 1.25 * P->seed_ratio_hint) : 6.0;
    const double want = per_base * (double)T->total;
    // u32 indices; 32 B of work list + record per seed (a 288 GB part holds the
    // 1.2e9 seeds of S4's largest group, 38 GB, without a second pass)
    // (a first scan without a hint is held to 2.5e9 seeds -- list, records and ranks of a seed take 68 B
    // until the rows are built, 170 GB at that size; a group that needs more pays the second scan once)
    const double cap = P->seed_ratio_hint > 0.0 ? 4.0e9 : 2.5e9;
    return (u32)std::max<double>((double)((i64)1 << 20), std::min<double>(want, cap));
}

// K1c host side: hash table of the anchor k-mers, one lookup per target
// position, one exact verification per seed.  Everything is stream-ordered;
// the number of seeds stays on the device (S.ctr[1]) and the hits go straight
// into the bucketed row build as records indexed like the seeds.
struct SeedRun {
    DevBuf<uint4> slot;
    DevBuf<u32> aslot;       // filtered look-up: the table slot of every anchor's k-mer
    DevBuf<u32> present;     // presence bits of the table's keys
    DevBuf<uint4> sib;
    DevBuf<uint2> ranges;    // filtered look-up: per look-up workgroup (first list entry, seeds kept)
    u32 nranges = 0;         // 0: unfiltered
    DevBuf<u32> cnt, ents, slot_of, ctr, spos, sent, sseq, dummy;
    u32 scap = 0;
};

// table of the anchor k-mers + one lookup per target position -> seed work
// list (S.spos / S.sent / S.sseq, count in S.ctr[1]).  pos_limit: anchors at or
// beyond this probe offset stay out of the table.  bcnt/nb/res: arrays of the
// bucketed row build zeroed by the same init launch (may be null / 0).
static int seed_table_lookup_async(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T, u32 pos_limit,
                                   int mm, bool allow_filter, SeedRun &S, u32 *bcnt, u32 nb, u32 *res, PhaseTimer &tm) {
    const int k = P->k, kb = std::min(k, 32);
    const u64 nent64 = (u64)P->nent;
    if (nent64 >= ((u64)1 << 31)) { chip_set_error("seed scan: too many anchors"); return CATCHHIP_EINVAL; }
    const u32 nent = (u32)nent64;
    u32 capacity = 1024;
    while ((u64)capacity < 2 * nent64) capacity <<= 1;
    TRY(S.slot.reserve(capacity));
    TRY(S.cnt.reserve(capacity));
    TRY(S.ents.reserve(nent));
    TRY(S.slot_of.reserve(nent));
    TRY(S.ctr.reserve(4));   // [0] ents cursor, [1] entries of the work list, [2] of them SEED_DEAD
    // the anchor-pair filter of the look-up (seed_lookup_kernel): pigeonhole tables of <= 4 anchors per probe
    // whose keys are whole k-mers with room for the N flags
    const int nanch_tab = P->pigeonhole ? (int)(P->L / k) : 0;
    // ... and (round 6) any other table of equal-length probes -- the reference's random anchors, -m 5: a pair is reported
    // from its LOWEST exact anchor, so a match whose probe has an exact anchor just below in the same window is not a seed
    const bool filt_random = !P->pigeonhole && P->L > 0 && (i64)P->L - k <= SL_HALO && P->L - k < 256 &&
                             !chip_test_env("CATCHHIP_SEED_RANDOM_KEEP_ALL");
    const bool filt = allow_filter && k <= 30 && !chip_test_env("CATCHHIP_SEED_KEEP_ALL") &&
                      (P->pigeonhole ? nanch_tab >= 2 && nanch_tab <= SEED_SIB + 1 && pos_limit != 0xffffffffu && (nanch_tab - 1) * k <= SL_HALO
                                     : filt_random);
    const int need2 = filt && nanch_tab - mm >= 2 ? 1 : 0;
    const u32 nblk = (u32)div_up(T->total, SL_TILE);
    S.nranges = filt ? nblk : 0;
    if (filt) { TRY(S.aslot.reserve(nent)); TRY(S.sib.reserve((size_t)nent)); TRY(S.ranges.reserve(nblk)); }
    TRY(S.spos.reserve(S.scap));
    TRY(S.sent.reserve(S.scap));
    TRY(S.sseq.reserve(S.scap));
    if (!res) { TRY(S.dummy.reserve(8)); res = S.dummy.p; }
    // presence bits: for the pigeonhole tables of a whole-genome scan (most positions miss); 4 per slot
    const bool pres = P->pigeonhole && capacity >= (1u << 16) && capacity <= (1u << 29) && !chip_test_env("CATCHHIP_SEED_NO_PRESENCE");
    const u32 pwords = pres ? capacity / 8 : 0;      // 4 * capacity bits
    if (pres) TRY(S.present.reserve(pwords));
    SeedTable t = {S.slot.p, S.cnt.p, S.ents.p, capacity - 1, filt ? (const uint4 *)S.sib.p : nullptr,
                   pres ? S.present.p : (u32 *)nullptr, pres ? 4 * capacity - 1 : 0u};
    const dim3 eb((unsigned)div_up((i64)nent, 256)), tb(256);
    hipLaunchKernelGGL(seed_init_kernel, dim3((unsigned)std::min<i64>(div_up((i64)capacity, 256), 2048)), tb, 0,
                       ctx->stream, S.slot.p, S.cnt.p, capacity, S.ctr.p, bcnt, nb, res, pres ? S.present.p : (u32 *)nullptr, pwords);
    hipLaunchKernelGGL(seed_count_kernel, eb, tb, 0, ctx->stream, (const uint4 *)P->planes.p,
                       (const u32 *)P->sent_probe.p, (const u32 *)P->sent_pos.p, nent, pos_limit,
                       nanch_tab, k, (int)P->pwords, kb, t, S.slot_of.p, filt ? S.aslot.p : (u32 *)nullptr);
    hipLaunchKernelGGL(seed_alloc_kernel, dim3(capacity / SA_SLOTS), dim3(256), 0, ctx->stream, t, S.ctr.p);
    hipLaunchKernelGGL(seed_fill_kernel, eb, tb, 0, ctx->stream, nent, t, (const u32 *)S.slot_of.p, nanch_tab,
                       (const u32 *)S.aslot.p, filt ? S.sib.p : nullptr, (const u32 *)P->sent_probe.p, (const u32 *)P->sent_pos.p);
    hipLaunchKernelGGL(seed_lookup_kernel, dim3((unsigned)div_up(T->total, SL_TILE)), dim3(SL_THREADS), 0, ctx->stream,
                       (const u32 *)T->planes.p, T->nwords, (u32)T->total, (const u32 *)T->seq_off.p,
                       (u32)T->nseq, k, kb, nanch_tab, need2, T->has_n ? 1 : 0, t, S.spos.p, S.sent.p, S.sseq.p,
                       S.ctr.p + 1, S.scap, filt ? S.ranges.p : (uint2 *)nullptr);
    tm.launch(5);
    return 0;
}

static int run_seed_async(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T, int mm, SeedRun &S,
                          const HitSink &sink, u32 nb, u32 *res, PhaseTimer &tm) {
    const bool use_n = P->has_n || T->has_n;
    // Pigeonhole anchors {0,k,..,L-k}: a window with <= mm mismatches leaves at
    // least one of ANY mm+1 disjoint anchors exact, so the first mm+1 anchors are
    // all the table needs (fewer seeds to verify).  Any other anchor table (the
    // reference's random anchors) enters completely: a pair is reported iff one
    // of ITS anchors matches exactly, as the reference finds it.
    const int k = P->k;
    const u32 pos_limit = P->pigeonhole ? (u32)std::min<i64>((i64)(mm + 1) * k, P->L) : 0xffffffffu;
    seed_verify_fn verify = pick_seed_verify((int)P->pwords);
    if (!verify) { chip_set_error("seed scan: unsupported probe length"); return CATCHHIP_EINVAL; }
    seed_verify4_fn verify4 = pick_seed_verify4((int)P->pwords);
    // anchors per probe in the table (pigeonhole: those below pos_limit), <= 31 for the bit set
    const int nanch = P->pigeonhole ? (int)(P->L / k) : 0;
    const int ntab = nanch ? (int)std::min<i64>(nanch, div_up((i64)pos_limit, k)) : 0;
    if (ntab > 31) verify4 = nullptr;
    // (the filtered look-up hands its seeds over as ranges, which only the cooperative verify kernel walks,
    // and it needs the hits filed compactly: a group of 64 list entries outside the ranges is never visited)
    TRY(seed_table_lookup_async(ctx, P, T, pos_limit, mm, verify4 != nullptr && sink.wcnt != nullptr, S, sink.bcnt, nb, res,
                                tm));
    if (S.nranges) HIP_TRY(hipMemsetAsync(sink.wcnt, 0, sizeof(u32) * ((size_t)S.scap / 64 + 1), ctx->stream));
    const u32 tailmask = (P->L & 31) ? ((1u << (P->L & 31)) - 1u) : 0xffffffffu;
    // the verify launch alone is phase 5 (read lazily by catchhip_ctx_last_kernel_ms)
    (void)hipEventRecord(ctx->ev[2 * PHASE_VERIFY], ctx->stream);
    if (verify4)
        hipLaunchKernelGGL(verify4, dim3(S.nranges ? S.nranges : (unsigned)div_up((i64)S.scap, 256)), dim3(256), 0, ctx->stream,
                           (const uint4 *)T->tq.p, (const u32 *)T->seq_off.p, (const uint4 *)P->planes.p,
                           (const u32 *)P->sent_probe.p, (const u32 *)P->sent_pos.p, (const u32 *)P->ent_ptr.p,
                           nanch, ntab, (int)P->L, k, mm, tailmask, use_n ? 1 : 0,
                           (const u32 *)S.spos.p, (const u32 *)S.sent.p, (const u32 *)S.sseq.p,
                           (const u32 *)(S.ctr.p + 1), S.scap, sink, S.nranges ? (const uint2 *)S.ranges.p : (const uint2 *)nullptr);
    else
    hipLaunchKernelGGL(verify, dim3((unsigned)div_up((i64)S.scap, 256)), dim3(256), 0, ctx->stream,
                       (const u32 *)T->planes.p, T->nwords, (const u32 *)T->seq_off.p,
                       (const uint4 *)P->planes.p, (const u32 *)P->sent_probe.p, (const u32 *)P->sent_pos.p,
                       (const u32 *)P->ent_ptr.p, P->pigeonhole ? (int)(P->L / k) : 0, (int)P->L, k, mm, tailmask,
                       use_n ? 1 : 0,
                       (const u32 *)S.spos.p, (const u32 *)S.sent.p, (const u32 *)S.sseq.p,
                       (const u32 *)(S.ctr.p + 1), S.scap, sink);
    (void)hipEventRecord(ctx->ev[2 * PHASE_VERIFY + 1], ctx->stream);
    ctx->phase_launches[PHASE_VERIFY] = 1;
    ctx->phase_launches[PHASE_VCOUNT] = 0; ctx->phase_ms[PHASE_VCOUNT] = 0.0;
    for (int q = 0; q < 4; ++q) ctx->join_counters[q] = 0;   // (no join in this scan)
    tm.launch(1);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int run_general(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T, int mm,
                       int lcf_thres, int island, RawHits &H, PhaseTimer &tm) {
    H.n = 0;
    H.has_end = true;
    if (P->nent == 0 || T->total == 0) return 0;
    if (mm > MAX_MM) { chip_set_error("mismatches > %d not supported", MAX_MM); return CATCHHIP_EINVAL; }
    const u32 nent = (u32)P->nent;
    // Seeds = exact matches of anchor k-mers.  Equal-length DNA probes have a
    // packed image: their anchors go into the hash table of the seed scan and
    // every target position does one look-up (the extension below re-checks the
    // k-mer on the bytes).  Anything else is joined through a sort of byte hashes.
    const bool table = P->dna5 && T->dna5 && P->L > 0 && P->L <= 256 && P->pwords >= 1 && P->k <= P->L &&
                       !chip_test_env("CATCHHIP_GENERAL_SORTJOIN");
    SeedRun S;
    DevBuf<u64> keys, keys_alt;
    DevBuf<u32> vals, vals_alt, sa, sb, scount;
    const u32 *seed_ent = nullptr, *seed_pos = nullptr;
    const i32 *e_probe = P->ent_probe.p, *e_pos = P->ent_pos.p;
    u32 nseeds = 0;
    if (table) {
        S.scap = seed_capacity(P, T);
        for (int attempt = 0;; ++attempt) {
            TRY(seed_table_lookup_async(ctx, P, T, 0xffffffffu, 0, false, S, nullptr, 0, nullptr, tm));
            HIP_TRY(hipGetLastError());
            TRY(read_count(ctx, S.ctr.p + 1, &nseeds));
            if (nseeds <= S.scap) break;
            if (attempt >= 2) { chip_set_error("seed lookup: work list overflow"); return CATCHHIP_ENOMEM; }
            S.scap = nseeds;
        }
        seed_ent = S.sent.p; seed_pos = S.spos.p;
        // the table's entries are the anchors sorted by (probe, position)
        e_probe = (const i32 *)P->sent_probe.p; e_pos = (const i32 *)P->sent_pos.p;
    } else {
        TRY(keys.alloc(nent));
        TRY(vals.alloc(nent));
        hipLaunchKernelGGL(anchor_hash_kernel, dim3((unsigned)div_up(nent, 256)), dim3(256), 0, ctx->stream,
                           P->bytes.p, P->probe_off.p, P->ent_probe.p, P->ent_pos.p, nent, (int)P->k, keys.p,
                           vals.p);
        tm.launch();
        TRY(chip_radix_sort_pairs(ctx, keys, keys_alt, vals, vals_alt, nent, 64));
        TRY(scount.alloc(1));
        u32 cap = (u32)std::max<i64>((i64)1 << 20, std::min<i64>(T->total * 2, (i64)1 << 28));
        for (int attempt = 0;; ++attempt) {
            TRY(sa.reserve(cap));
            TRY(sb.reserve(cap));
            HIP_TRY(hipMemsetAsync(scount.p, 0, sizeof(u32), ctx->stream));
            HitBuf hb = {sa.p, sb.p, nullptr, scount.p, cap};
            hipLaunchKernelGGL(seed_join_kernel, dim3((unsigned)div_up(T->total, 256)), dim3(256), 0, ctx->stream,
                               T->bytes.p, (u32)T->total, T->seq_off.p, (u32)T->nseq, (int)P->k, keys.p, vals.p,
                               nent, hb);
            tm.launch();
            HIP_TRY(hipGetLastError());
            TRY(read_count(ctx, scount.p, &nseeds));
            if (nseeds <= cap) break;
            if (attempt >= 2) { chip_set_error("seed join: hit buffer overflow"); return CATCHHIP_ENOMEM; }
            cap = nseeds;
        }
        seed_ent = sa.p; seed_pos = sb.p;
    }
    if (nseeds == 0) return 0;
    // extension: at most one range per seed
    TRY(H.a.reserve(nseeds));
    TRY(H.b.reserve(nseeds));
    TRY(H.c.reserve(nseeds));
    TRY(H.count.alloc(1));
    HIP_TRY(hipMemsetAsync(H.count.p, 0, sizeof(u32), ctx->stream));
    if (H.want_seed) {
        TRY(H.d.reserve(nseeds));
        TRY(H.e.reserve(nseeds));
    }
    HitBuf ob = {H.a.p, H.b.p, H.c.p, H.count.p, nseeds, H.want_seed ? H.d.p : nullptr,
                 H.want_seed ? H.e.p : nullptr};
    extend_planes_fn planes = table && !chip_test_env("CATCHHIP_EXTEND_BYTES") ? pick_extend_planes((int)P->pwords) : nullptr;
    DevBuf<u32> cut;
    const u32 cut_cap = 1u << 22;
    if (planes) {
        // packed images: the window's mismatch mask in a few words, bit scans around the anchor
        TRY(cut.alloc((size_t)cut_cap + 1));
        HIP_TRY(hipMemsetAsync(cut.p, 0, sizeof(u32), ctx->stream));
        const u32 tailmask = (P->L & 31) ? ((1u << (P->L & 31)) - 1u) : 0xffffffffu;
        hipLaunchKernelGGL(planes, dim3((unsigned)div_up(nseeds, 256 * SE_PPT)), dim3(256), 0, ctx->stream,
                           (const u32 *)T->planes.p, T->nwords, (const u8 *)T->bytes.p, (const u32 *)T->seq_off.p,
                           (const uint4 *)P->planes.p, (const u8 *)P->bytes.p, (const u32 *)P->probe_off.p, e_probe,
                           e_pos, (int)P->L, (int)P->k, mm, lcf_thres, island, tailmask,
                           (P->has_n || T->has_n) ? 1 : 0, seed_ent, seed_pos, (const u32 *)S.sseq.p, nseeds, ob,
                           cut.p, cut_cap);
        hipLaunchKernelGGL(extend_cut_kernel, dim3(256), dim3(256), 0, ctx->stream, (const u32 *)cut.p, cut_cap,
                           (const u8 *)T->bytes.p, (const u32 *)T->seq_off.p, (const u8 *)P->bytes.p,
                           (const u32 *)P->probe_off.p, e_probe, e_pos, (int)P->k, mm, lcf_thres, island, seed_ent,
                           seed_pos, (const u32 *)S.sseq.p, ob);
        tm.launch();
        u32 ncut = 0;
        TRY(read_count(ctx, cut.p, &ncut));
        if (ncut > cut_cap) {   // never seen: more cut windows than the list holds -> everything by bytes
            HIP_TRY(hipMemsetAsync(H.count.p, 0, sizeof(u32), ctx->stream));
            planes = nullptr;
        }
    }
    if (!planes) {
        hipLaunchKernelGGL(extend_kernel, dim3((unsigned)div_up(nseeds, 256)), dim3(256), 0, ctx->stream, T->bytes.p,
                           T->seq_off.p, (u32)T->nseq, P->bytes.p, P->probe_off.p, e_probe, e_pos,
                           (int)P->k, mm, lcf_thres, island, seed_ent, seed_pos, nseeds, ob);
    }
    tm.launch();
    HIP_TRY(hipGetLastError());
    TRY(read_count(ctx, H.count.p, &H.n));
    ctx->counters[0] = H.n;
    ctx->counters[1] = nseeds;
    ctx->seeds_dropped = 0;
    return 0;
}

// ------------------------------------------------------------------------
// row build, host side
// ------------------------------------------------------------------------
struct BucketBuild {
    DevBuf<uint4> rec;
    DevBuf<uint4> S;   // hits grouped by bucket, then the merged rows, {start, end, segment, bucket}
    DevBuf<u32> rank, bcnt, bstart, mcnt, blmax, rstart, res, tsum, tamax, wcnt;
    bool compact = false;   // the producer files its hits compactly (HitSink::wcnt)
    DevBuf<unsigned long long> bsum;
    u32 nb = 0, cap = 0;
};
// res words: [0] unused, [1] overflow flag (a bucket beyond BK_BIG), [2] hits, [3] largest bucket,
//            [4] merged rows, [5] longest row   ([3] is no longer filled)

static int bucket_prepare(BucketBuild &B, u32 nb, u32 cap, bool want_sum) {
    B.nb = nb; B.cap = cap;
    TRY(B.rec.reserve(cap));
    TRY(B.rank.reserve(cap));
    TRY(B.wcnt.reserve((size_t)cap / 64 + 2));
    B.compact = false;
    TRY(B.S.reserve(cap));
    TRY(B.bcnt.reserve((size_t)nb + 1));
    TRY(B.bstart.reserve((size_t)nb + 2));
    TRY(B.mcnt.reserve((size_t)nb + 1));
    TRY(B.blmax.reserve((size_t)nb + 1));
    TRY(B.rstart.reserve((size_t)nb + 2));
    TRY(B.res.reserve(8));
    if (want_sum) TRY(B.bsum.reserve((size_t)nb + 1));
    return 0;
}

// exclusive scan of n counts (device-resident results): one workgroup for
// small inputs, tiles otherwise
static int bucket_scan(catchhip_ctx *ctx, BucketBuild &B, const u32 *in, u32 *out, u32 n, u32 *total_out,
                       const u32 *aux, u32 *auxmax_out, PhaseTimer &tm) {
    hipStream_t s = ctx->stream;
    if (n <= 16384) {
        static const int s1t = chip_test_env("CATCHHIP_SCAN1_THREADS") ? atoi(chip_test_env("CATCHHIP_SCAN1_THREADS")) : 1024;
        hipLaunchKernelGGL(scan1_kernel, dim3(1), dim3(s1t), 0, s, in, out, n, total_out, (u32 *)nullptr, aux,
                           auxmax_out);
        tm.launch(1);
        return 0;
    }
    const u32 ntiles = (u32)div_up((i64)n, SCT_TILE);
    TRY(B.tsum.reserve(ntiles));
    TRY(B.tamax.reserve(ntiles));
    hipLaunchKernelGGL(scan_tiles_reduce_kernel, dim3(ntiles), dim3(SCT_THREADS), 0, s, in, n, aux, B.tsum.p,
                       B.tamax.p);
    hipLaunchKernelGGL(scan_tiles_apply_kernel, dim3(ntiles), dim3(SCT_THREADS), 0, s, in, out, n,
                       (const u32 *)B.tsum.p, (const u32 *)B.tamax.p, ntiles, total_out, auxmax_out);
    tm.launch(2);
    return 0;
}

// scan of the bucket sizes, scatter, per-bucket sort + merge, scan of the
// merged counts.  nrec = hit records to look at (a device count, bounded by
// B.cap, when nrec_dev is given).
// grouped: the producer (run_join) has scanned the bucket sizes and written its records at their places in B.S.
static int bucket_finish_async(catchhip_ctx *ctx, BucketBuild &B, u32 nrec, const u32 *nrec_dev, bool want_sum,
                               bool merge, PhaseTimer &tm, bool dedupe = false, bool grouped = false,
                               const u32 *run_cnt = nullptr, int run_stride = 0, int nruns = 0) {
    hipStream_t s = ctx->stream;
    if (!grouped) TRY(bucket_scan(ctx, B, B.bcnt.p, B.bstart.p, B.nb, B.res.p + 2, nullptr, nullptr, tm));
    if (!merge) return 0;   // radix build: only the bucket offsets are needed
    if (nrec && !grouped)
        hipLaunchKernelGGL(bucket_scatter_kernel, dim3((unsigned)div_up((i64)nrec, 256)), dim3(256), 0, s,
                           (const uint4 *)B.rec.p, (const u32 *)B.rank.p, nrec, nrec_dev, (const u32 *)B.bstart.p,
                           B.S.p, B.compact ? (const u32 *)B.wcnt.p : (const u32 *)nullptr);
    unsigned long long *bsum = want_sum ? B.bsum.p : nullptr;
    hipLaunchKernelGGL((bucket_merge_kernel<64, BK_SMALL>), dim3((unsigned)std::min<i64>(B.nb, (i64)1 << 20)), dim3(64),
                       0, s, (const u32 *)B.bstart.p, B.nb, B.S.p, B.mcnt.p, B.blmax.p, bsum,
                       dedupe ? 1 : 0, run_cnt, run_stride, nruns);
    static bool big_attr_set = false;
    if (!big_attr_set) {
        (void)hipFuncSetAttribute((const void *)bucket_merge_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  3 * BK_BIG * (int)sizeof(u32));
        big_attr_set = true;
    }
    // four size classes, each with LDS arrays of its own length: 12 / 24 / 48 / 96 KB = 4 / 4 / 3 / 1 workgroups per CU
    {
        u32 lo = BK_SMALL;
        for (u32 cap = 1024; cap <= (u32)BK_BIG; cap <<= 1) {
            const unsigned per_cu = cap <= 2048 ? 4u : cap <= 4096 ? 3u : 1u;
            hipLaunchKernelGGL(bucket_merge_big_kernel, dim3((unsigned)ctx->num_cus * per_cu), dim3(BKB_THREADS), 3 * cap * sizeof(u32), s,
                               (const u32 *)B.bstart.p, B.nb, B.S.p, B.mcnt.p, B.blmax.p, bsum,
                               B.res.p + 1, dedupe ? 1 : 0, lo, cap);
            lo = cap;
        }
    }
    tm.launch(6);
    TRY(bucket_scan(ctx, B, B.mcnt.p, B.rstart.p, B.nb, B.res.p + 4, B.blmax.p, B.res.p + 5, tm));
    return 0;
}

// ------------------------------------------------------------------------
// K1d host side (scan_join.inc): table -> hit positions -> sort by slot -> count pass -> bucket offsets ->
// write pass.  Leaves the hit records grouped by bucket in B.S with B.bstart / B.res[2] filled, i.e. where
// bucket_finish_async's scatter would have left them.  Two host synchronisations (hit positions, hits): the
// arrays are sized exactly.  Returns 1 when the inputs do not qualify (the caller takes the seed-list scan).
// ------------------------------------------------------------------------
struct JoinRun {
    DevBuf<u64> stage_key, hkey, hkey_alt;
    DevBuf<u32> stage_sq, hsq, hsq_alt, wg_cnt, wg_off, scan_tmp, ecnt, ebase, bcur, giant_n;
    DevBuf<uint4> giant;
    DevBuf<unsigned long long> pairs, masks;
    DevBuf<u32> mbase, gmbase;
    JoinArgs A;
    kj_kernel_fn write_main = nullptr, write_giant = nullptr;
    u32 nhit = 0, ngiant = 0;
    bool used = false;
};
#define KJ_GIANT_CAP (1u << 20)

__global__ void __launch_bounds__(256)
rows_gain0_kernel(const unsigned long long *__restrict__ bsum, u32 nb, const i32 *__restrict__ bucket_set, u32 ng,
                  u32 *__restrict__ gain0) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const u32 v = (u32)bsum[b];
    const u32 s = bucket_set ? (u32)bucket_set[b] : b;
    if (v && s < ng) atomicAdd(&gain0[s], v);
}

static bool join_path_ok(const catchhip_probes *P, int mm) {
    if (!P->pigeonhole || P->k <= 0 || P->pwords < 1 || P->pwords > 8) return false;
    const int nanch = (int)(P->L / P->k);
    const int ntab = (int)std::min<i64>(nanch, (i64)mm + 1);
    return ntab >= 1 && ntab <= KJ_AMAX && !chip_test_env("CATCHHIP_SEED_LIST");
}

// the write pass alone (again after a row build that fell back to the radix sort: the merge works in place)
static void join_write_pass(catchhip_ctx *ctx, JoinRun &J) {
    if (!J.nhit) return;
    hipLaunchKernelGGL(J.write_main, dim3((unsigned)div_up((i64)J.nhit, 256)), dim3(256), 0, ctx->stream, J.A);
    if (J.ngiant) (void)hipMemsetAsync(J.giant_n.p + 2, 0, sizeof(u32), ctx->stream);   // the write pass's task cursor
    if (J.ngiant)
        hipLaunchKernelGGL(J.write_giant, dim3((unsigned)std::min<i64>(div_up((i64)J.ngiant, 4), (i64)ctx->num_cus * 8)),
                           dim3(256), 0, ctx->stream, J.A);
}

static int run_join(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T, int mm, SeedRun &S, JoinRun &J,
                    const HitSink &sink, BucketBuild &B, u32 nb, PhaseTimer &tm) {
    hipStream_t s = ctx->stream;
    const int k = P->k, kb = std::min(k, 32), NW = (int)P->pwords;
    const int nanch = (int)(P->L / k);
    const u32 pos_limit = (u32)std::min<i64>((i64)(mm + 1) * k, P->L);
    const int ntab = (int)std::min<i64>(nanch, div_up((i64)pos_limit, k));
    const u64 nent64 = (u64)P->nent;
    if (nent64 >= ((u64)1 << 31)) { chip_set_error("seed scan: too many anchors"); return CATCHHIP_EINVAL; }
    const u32 nent = (u32)nent64;
    // ---- table of the anchor k-mers (as the seed-list scan builds it, without the sibling records) ----------
    u32 capacity = 1024;
    while ((u64)capacity < 2 * nent64) capacity <<= 1;
    TRY(S.slot.reserve(capacity));
    TRY(S.cnt.reserve(capacity));
    TRY(S.ents.reserve(nent));
    TRY(S.slot_of.reserve(nent));
    TRY(S.ctr.reserve(4));
    const bool pres = capacity >= (1u << 16) && capacity <= (1u << 29) && !chip_test_env("CATCHHIP_SEED_NO_PRESENCE");
    // One presence bit per slot (4 MB for the 32 M slots of S4's largest group: mostly L2 hits; the seed-list scan's
    // 4 bits per slot are 16 MB: every probe a trip to the memory-side cache).  Measured on S4, whole scan phase:
    // 4 bits 33.8 ms, 1 bit 30.5, half a bit 30.4, a quarter 30.8 (more false positives go on to the slot array).
    static const int pshift = chip_test_env("CATCHHIP_PRESENCE_SHIFT") ? atoi(chip_test_env("CATCHHIP_PRESENCE_SHIFT")) : 2;
    const u32 pbits = pres ? std::max<u32>((4u * capacity) >> pshift, 1u << 16) : 0;
    const u32 pwords = pbits / 32;
    if (pres) TRY(S.present.reserve(pwords));
    SeedTable t = {S.slot.p, S.cnt.p, S.ents.p, capacity - 1, nullptr, pres ? S.present.p : (u32 *)nullptr,
                   pres ? pbits - 1 : 0u};
    const dim3 eb((unsigned)div_up((i64)nent, 256)), tb(256);
    hipLaunchKernelGGL(seed_init_kernel, dim3((unsigned)std::min<i64>(div_up((i64)capacity, 256), 2048)), tb, 0, s,
                       S.slot.p, S.cnt.p, capacity, S.ctr.p, B.bcnt.p, nb, B.res.p, pres ? S.present.p : (u32 *)nullptr, pwords);
    hipLaunchKernelGGL(seed_count_kernel, eb, tb, 0, s, (const uint4 *)P->planes.p, (const u32 *)P->sent_probe.p,
                       (const u32 *)P->sent_pos.p, nent, pos_limit, nanch, k, NW, kb, t, S.slot_of.p, (u32 *)nullptr);
    hipLaunchKernelGGL(seed_alloc_kernel, dim3(capacity / SA_SLOTS), dim3(256), 0, s, t, S.ctr.p);
    hipLaunchKernelGGL(seed_fill_kernel, eb, tb, 0, s, nent, t, (const u32 *)S.slot_of.p, nanch, (const u32 *)nullptr,
                       (uint4 *)nullptr, (const u32 *)nullptr, (const u32 *)nullptr);
    // ---- hit positions, in position order ------------------------------------------------------------------
    const u32 nblk = (u32)div_up(T->total, KJ_TILE);
    TRY(J.stage_key.reserve((size_t)nblk * KJ_TILE));
    TRY(J.stage_sq.reserve((size_t)nblk * KJ_TILE));
    TRY(J.wg_cnt.reserve((size_t)nblk + 1));
    TRY(J.wg_off.reserve((size_t)nblk + 1));
    hipLaunchKernelGGL(kj_hitpos_kernel, dim3(nblk), dim3(KJ_HT), 0, s, (const u32 *)T->planes.p, T->nwords, (u32)T->total,
                       (const u32 *)T->seq_off.p, (u32)T->nseq, k, kb, t, (unsigned long long *)J.stage_key.p, J.stage_sq.p,
                       J.wg_cnt.p);
    TRY(chip_exclusive_scan_u32(ctx, J.wg_cnt.p, J.wg_off.p, (i64)nblk + 1, J.scan_tmp));
    tm.launch(7);
    TRY(read_count(ctx, J.wg_off.p + nblk, &J.nhit));
    J.used = true;
    J.ngiant = 0;
    TRY(J.ecnt.reserve((size_t)nent + 1));
    TRY(J.ebase.reserve((size_t)nent + 1));
    TRY(J.giant.reserve(KJ_GIANT_CAP));
    TRY(J.giant_n.reserve(4));
    TRY(J.pairs.reserve(192));   // [0..128) statistics (sharded), [128..192) the mask store's cursors
    HIP_TRY(hipMemsetAsync(J.ecnt.p, 0, sizeof(u32) * ((size_t)nent + 1), s));
    HIP_TRY(hipMemsetAsync(J.giant_n.p, 0, sizeof(u32) * 4, s));
    HIP_TRY(hipMemsetAsync(J.pairs.p, 0, sizeof(unsigned long long) * 192, s));
    const bool cursor = sink.bucket_of != nullptr;   // several probes may share a bucket
    if (cursor) { TRY(J.bcur.reserve((size_t)nb + 1)); HIP_TRY(hipMemsetAsync(J.bcur.p, 0, sizeof(u32) * ((size_t)nb + 1), s)); }
    JoinArgs &A = J.A;
    A.tq = (const uint4 *)T->tq.p; A.seq_off = (const u32 *)T->seq_off.p; A.pplanes = (const uint4 *)P->planes.p;
    A.nanch = nanch; A.ntab = ntab; A.L = (int)P->L; A.k = k; A.mm = mm;
    A.div_magic = (((unsigned long long)1 << 34) + (unsigned long long)nanch - 1ull) / (unsigned long long)nanch;
    A.tailmask = (P->L & 31) ? ((1u << (P->L & 31)) - 1u) : 0xffffffffu;
    A.nhit = J.nhit;
    A.slot = (const uint4 *)S.slot.p; A.ents = (const u32 *)S.ents.p;
    A.ecnt = J.ecnt.p; A.ebase = (const u32 *)J.ebase.p; A.S = nullptr;
    A.bucket_of = sink.bucket_of; A.seq_genome = sink.seq_genome; A.ext = sink.ext;
    A.probe_group = sink.probe_group; A.seq_group = sink.seq_group;
    A.giant = J.giant.p; A.giant_n = J.giant_n.p; A.giant_cap = KJ_GIANT_CAP;
    A.giant_pairs = chip_test_env("CATCHHIP_JOIN_GIANT_PAIRS") ? (u32)atoi(chip_test_env("CATCHHIP_JOIN_GIANT_PAIRS")) : KJ_GIANT_PAIRS;
    // hit masks of the counting pass for the writing pass (scan_join.inc): 6 words per hit position hold S4's
    // (2 per position there); a run that finds the store full is simply verified again
    A.masks = nullptr; A.mbase = A.gmbase = nullptr; A.mcursor = J.pairs.p + 128; A.mask_cap = 0;   // (zeroed with the statistics)
    if (J.nhit && !chip_test_env("CATCHHIP_JOIN_NO_MASKS")) {
        const size_t mcap = (size_t)std::min<u64>((u64)1 << 26, std::max<u64>((u64)1 << 20, 6ull * J.nhit));
        TRY(J.masks.reserve(mcap));
        TRY(J.mbase.reserve(J.nhit));
        TRY(J.gmbase.reserve(KJ_GIANT_CAP));
        A.masks = J.masks.p; A.mbase = J.mbase.p; A.gmbase = J.gmbase.p; A.mask_cap = (u32)mcap;
    }
    A.pairs = J.pairs.p;
    J.write_main = pick_kj_verify<true>(NW);
    J.write_giant = pick_kj_giant<true>(NW);
    ctx->phase_launches[PHASE_VCOUNT] = 0; ctx->phase_ms[PHASE_VCOUNT] = 0.0;
    if (J.nhit) {
        TRY(J.hkey.reserve(J.nhit));
        TRY(J.hsq.reserve(J.nhit));
        hipLaunchKernelGGL(kj_compact_kernel, dim3(nblk), dim3(64), 0, s, (const unsigned long long *)J.stage_key.p,
                           (const u32 *)J.stage_sq.p, (const u32 *)J.wg_off.p, (unsigned long long *)J.hkey.p, J.hsq.p);
        // stable: the records of a slot stay in position order
        TRY(chip_radix_sort_pairs(ctx, J.hkey, J.hkey_alt, J.hsq, J.hsq_alt, (i64)J.nhit, ceil_log2_u64((u64)capacity), 32));
        A.hkey = (const unsigned long long *)J.hkey.p; A.hsq = (const u32 *)J.hsq.p;
        (void)hipEventRecord(ctx->ev[2 * PHASE_VCOUNT], s);
        hipLaunchKernelGGL(pick_kj_verify<false>(NW), dim3((unsigned)div_up((i64)J.nhit, 256)), dim3(256), 0, s, A);
        hipLaunchKernelGGL(pick_kj_giant<false>(NW), dim3((unsigned)ctx->num_cus * 2), dim3(256), 0, s, A);
        (void)hipEventRecord(ctx->ev[2 * PHASE_VCOUNT + 1], s);
        ctx->phase_launches[PHASE_VCOUNT] = 1;
        tm.launch(3 + 5 * ((ceil_log2_u64((u64)capacity) + 7) / 8));
    }
    hipLaunchKernelGGL(kj_bucket_count_kernel, dim3((unsigned)div_up(P->nprobes, 256)), dim3(256), 0, s, (const u32 *)J.ecnt.p,
                       (u32)P->nprobes, nanch, ntab, sink.bucket_of, B.bcnt.p);
    tm.launch(1);
    TRY(bucket_scan(ctx, B, B.bcnt.p, B.bstart.p, nb, B.res.p + 2, nullptr, nullptr, tm));
    HIP_TRY(hipGetLastError());
    u32 *h = (u32 *)ctx->h_pin;
    HIP_TRY(hipMemcpyAsync(h, B.res.p + 2, sizeof(u32), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h + 1, J.giant_n.p, sizeof(u32), hipMemcpyDeviceToHost, s));
    TRY(chip_pinned_reserve(ctx, sizeof(unsigned long long) * 128));
    unsigned long long *hp = (unsigned long long *)ctx->h_big;
    HIP_TRY(hipMemcpyAsync(hp, J.pairs.p, sizeof(unsigned long long) * 128, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const u32 nhits = ((volatile u32 *)h)[0];
    J.ngiant = ((volatile u32 *)h)[1];
    unsigned long long pairs = 0, slots = 0;
    for (int i = 0; i < 64; ++i) { pairs += ((volatile unsigned long long *)hp)[i]; slots += ((volatile unsigned long long *)hp)[64 + i]; }
    if (getenv("CATCHHIP_TIMING"))
        fprintf(stderr, "[catchhip]   join: %u hit positions, %llu pairs, %llu lane slots in wave-wide runs, %u hits, %u tasks of cut runs\n",
                J.nhit, pairs, slots, nhits, J.ngiant);
    if (J.ngiant > KJ_GIANT_CAP) return 1;   // absurdly repetitive input: the seed-list scan copes (slowly)
    ctx->counters[1] = (i64)pairs;
    ctx->seeds_dropped = 0;
    ctx->join_counters[0] = J.nhit; ctx->join_counters[1] = (i64)pairs; ctx->join_counters[2] = (i64)slots; ctx->join_counters[3] = J.ngiant;
    TRY(B.S.reserve(std::max<size_t>(nhits, 1)));
    A.S = B.S.p;
    hipLaunchKernelGGL(kj_bases_kernel, dim3((unsigned)div_up(P->nprobes, 256)), dim3(256), 0, s, (const u32 *)J.ecnt.p,
                       (u32)P->nprobes, nanch, ntab, sink.bucket_of, (const u32 *)B.bstart.p, cursor ? J.bcur.p : (u32 *)nullptr,
                       J.ebase.p);
    (void)hipEventRecord(ctx->ev[2 * PHASE_VERIFY], s);
    join_write_pass(ctx, J);
    (void)hipEventRecord(ctx->ev[2 * PHASE_VERIFY + 1], s);
    ctx->phase_launches[PHASE_VERIFY] = 1;
    tm.launch(2 + (J.ngiant ? 1 : 0));
    HIP_TRY(hipGetLastError());
    return 0;
}

// radix-sort build (fallback for buckets beyond BK_BIG / very many buckets):
// keys (set id << 32 | start) + ends from the hit records
__global__ void __launch_bounds__(256)
rec_keys_kernel(const uint4 *__restrict__ rec, const u32 *__restrict__ rank, u32 nrec_cap,
                const u32 *__restrict__ nrec_dev, const u32 *__restrict__ bstart,
                const i32 *__restrict__ bucket_set, u64 *__restrict__ keys, u32 *__restrict__ vals,
                const u32 *__restrict__ wcnt) {
    const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 n = nrec_dev ? min(*nrec_dev, nrec_cap) : nrec_cap;
    if (d >= n) return;
    if (wcnt && (d & 63u) >= wcnt[d >> 6]) return;
    const u32 rk = rank[d];
    if (rk == BK_NONE) return;
    const uint4 r = rec[d];
    const u32 slot = bstart[r.w] + rk;   // the bucket offsets give every hit its own slot
    keys[slot] = ((u64)(bucket_set ? (u32)bucket_set[r.w] : r.w) << 32) | r.x;
    vals[slot] = r.y;
}

struct MergedRows {
    DevBuf<u64> keys, keys_alt;
    DevBuf<u32> vals, vals_alt, head, mend, seg, idx, tmp;
    u32 n = 0;       // sorted raw rows
    u32 nmerged = 0; // merged rows
};

// the same from records that are already grouped (key-grouped join: B.S, written again by its write pass)
__global__ void __launch_bounds__(256)
rec_keys_grouped_kernel(const uint4 *__restrict__ S, u32 n, const i32 *__restrict__ bucket_set, u64 *__restrict__ keys,
                        u32 *__restrict__ vals) {
    const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n) return;
    const uint4 r = S[d];
    keys[d] = ((u64)(bucket_set ? (u32)bucket_set[r.w] : r.w) << 32) | r.x;
    vals[d] = r.y;
}

static int build_rows_radix(catchhip_ctx *ctx, const BucketBuild &B, u32 nrec, const u32 *nrec_dev, u32 nhits,
                            const i32 *bucket_set, const u32 *bounds, u32 nbounds, i64 max_set_id, MergedRows &M,
                            PhaseTimer &tm, JoinRun *J = nullptr) {
    M.n = nhits;
    M.nmerged = 0;
    if (nhits == 0) return 0;
    const u32 n = nhits;
    TRY(M.keys.alloc(n));
    TRY(M.vals.alloc(n));
    if (J) {
        // the bucket merge works in place: have the join write its records again, then key them
        join_write_pass(ctx, *J);
        hipLaunchKernelGGL(rec_keys_grouped_kernel, dim3((unsigned)div_up((i64)n, 256)), dim3(256), 0, ctx->stream,
                           (const uint4 *)B.S.p, n, bucket_set, M.keys.p, M.vals.p);
        tm.launch(2);
    } else {
    hipLaunchKernelGGL(rec_keys_kernel, dim3((unsigned)div_up((i64)nrec, 256)), dim3(256), 0, ctx->stream,
                       (const uint4 *)B.rec.p, (const u32 *)B.rank.p, nrec, nrec_dev, (const u32 *)B.bstart.p,
                       bucket_set, M.keys.p, M.vals.p, B.compact ? (const u32 *)B.wcnt.p : (const u32 *)nullptr);
    tm.launch();
    }
    int bits = 32 + ceil_log2_u64((u64)max_set_id + 1);
    if (bits > 64) bits = 64;
    TRY(chip_radix_sort_pairs(ctx, M.keys, M.keys_alt, M.vals, M.vals_alt, n, bits));
    tm.launch(3 * ((bits + 7) / 8));
    TRY(M.head.alloc(n));
    TRY(M.mend.alloc(n));
    TRY(M.seg.alloc(n));
    TRY(M.idx.alloc(n));
    HIP_TRY(hipMemsetAsync(M.head.p, 0, sizeof(u32) * n, ctx->stream));
    hipLaunchKernelGGL(rows_merge_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, ctx->stream, M.keys.p,
                       M.vals.p, n, bounds, nbounds, M.head.p, M.mend.p, M.seg.p);
    TRY(chip_exclusive_scan_u32(ctx, M.head.p, M.idx.p, n, M.tmp));
    tm.launch(3);
    // merged count = idx[n-1] + head[n-1]
    HIP_TRY(hipMemcpyAsync(ctx->h_pin, M.idx.p + (n - 1), sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync((u32 *)ctx->h_pin + 1, M.head.p + (n - 1), sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    M.nmerged = ((volatile u32 *)ctx->h_pin)[0] + ((volatile u32 *)ctx->h_pin)[1];
    return 0;
}

// Scan + grouped hit records for either consumer.  by_sequence: merge per
// (probe, sequence) with buckets = probes (tolerant bp); otherwise per (set,
// genome) with buckets = set ids and the cover extension applied.
struct ScanOut {
    BucketBuild B;
    SeedRun S;
    RawHits H;                 // general path (kept for the first-seen pass)
    bool from_seeds = false;   // records come from the seed work list (S), not from H
    JoinRun J;                 // key-grouped join: hit list + what the write pass needs to run again
    bool from_join = false;    // records were written grouped (B.S) by the join; there is no rec / rank
    u32 nrec = 0;              // records to look at (grid size)
    const u32 *nrec_dev = nullptr;
    u32 nhits = 0, nrows = 0, lmax = 0, maxbucket = 0;
    bool overflow = false;     // a bucket beyond BK_BIG: use the radix build
};

static int scan_and_group(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T, int mismatches,
                          int lcf_thres, int island, u32 ext, bool by_sequence, int mode, ScanOut &O,
                          bool dedupe = false, bool want_first = false) {
    const bool seed_ok = seed_path_ok(P, T, mismatches, lcf_thres, island);
    const bool tiled_ok = fast_path_ok(P, T, mismatches, lcf_thres, island);
    const bool want_tiled = mode == CATCHHIP_SCAN_FAST || (mode == CATCHHIP_SCAN_AUTO && chip_test_env("CATCHHIP_SCAN_TILED"));
    const bool use_fast = tiled_ok && want_tiled && !(want_first && mode == CATCHHIP_SCAN_AUTO);
    if (use_fast && want_first) {
        chip_set_error("cover_scan_first_seen: the tiled scan does not know which anchor seeded a hit");
        return CATCHHIP_EINVAL;
    }
    O.from_seeds = false;
    const bool use_seed = seed_ok && !use_fast && mode != CATCHHIP_SCAN_GENERAL && mode != CATCHHIP_SCAN_FAST;
    const u32 nb = by_sequence ? (u32)P->nprobes : (u32)P->nbuckets;
    const bool force_radix = chip_test_env("CATCHHIP_ROWS_RADIX") != nullptr && !dedupe;
    HitSink sink;
    sink.bucket_of = by_sequence || P->bucket_identity ? nullptr : P->bucket_of.p;   // (null: the probe index itself)
    sink.seq_genome = by_sequence ? nullptr : T->seq_genome.p;
    sink.ext = ext;
    if (!by_sequence) {
        if (P->has_groups != T->has_groups) {
            chip_set_error("scan: groups must be set on both the probes and the targets, or on neither");
            return CATCHHIP_EINVAL;
        }
        if (P->has_groups) { sink.probe_group = P->group.p; sink.seq_group = T->seq_group.p; }
    }
    PhaseTimer ts(ctx, PHASE_SCAN), tr(ctx, PHASE_ROWS);
    bool use_join = use_seed && !want_first && join_path_ok(P, mismatches);
    if (use_join) {
        // key-grouped join (scan_join.inc): leaves the records grouped by bucket in O.B.S
        TRY(bucket_prepare(O.B, nb, 1, true));
        sink.rec = nullptr; sink.rank = nullptr; sink.bcnt = O.B.bcnt.p; sink.wcnt = nullptr;
        ts.restart();
        const int rc = run_join(ctx, P, T, mismatches, O.S, O.J, sink, O.B, nb, ts);
        if (rc < 0) return rc;
        if (rc > 0) use_join = false;       // did not qualify after all: the seed-list scan below
        else {
            ts.stop();
            O.from_join = true;
            O.nrec = 0; O.nrec_dev = nullptr;
            tr.restart();
            // (buckets = probes: every bucket is its probe's anchor runs side by side, each in position order)
            const bool runs = sink.bucket_of == nullptr && O.J.A.ntab <= BK_RUNS_MAX && !chip_test_env("CATCHHIP_MERGE_NO_RUNS");
            TRY(bucket_finish_async(ctx, O.B, 0, nullptr, true, !force_radix, tr, dedupe, true,
                                    runs ? (const u32 *)O.J.ecnt.p : (const u32 *)nullptr, O.J.A.nanch, O.J.A.ntab));
            tr.stop();
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(ctx->h_pin, O.B.res.p, 8 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
    }
    if (use_join) {
    } else if (use_seed) {
        O.from_seeds = true;
        O.S.scap = seed_capacity(P, T);
        if (const char *e = chip_test_env("CATCHHIP_SEED_CAP")) O.S.scap = (u32)std::max(1, atoi(e));   // tests: force the retry
        for (int attempt = 0;; ++attempt) {
            const auto dbg0 = std::chrono::steady_clock::now();
            TRY(bucket_prepare(O.B, nb, O.S.scap, true));
            sink.rec = O.B.rec.p; sink.rank = O.B.rank.p; sink.bcnt = O.B.bcnt.p;
            // (the first-discovery keys pair record d with seed d: not compact then)
            O.B.compact = !want_first && !chip_test_env("CATCHHIP_HITS_SPARSE");
            sink.wcnt = O.B.compact ? O.B.wcnt.p : nullptr;
            ts.restart();
            TRY(run_seed_async(ctx, P, T, mismatches, O.S, sink, nb, O.B.res.p, ts));
            if (getenv("CATCHHIP_TIMING"))
                fprintf(stderr, "[catchhip]   seed scan attempt %d: work list of %u seeds, %.3f ms to queue\n", attempt,
                        O.S.scap, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg0).count());
            ts.stop();
            O.nrec = O.S.scap; O.nrec_dev = O.S.ctr.p + 1;
            tr.restart();
            TRY(bucket_finish_async(ctx, O.B, O.nrec, O.nrec_dev, true, !force_radix, tr, dedupe));
            tr.stop();
            HIP_TRY(hipGetLastError());
            u32 *h = (u32 *)ctx->h_pin;
            HIP_TRY(hipMemcpyAsync(h, O.B.res.p, 8 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipMemcpyAsync(h + 8, O.S.ctr.p, 4 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            const u32 nseeds = ((volatile u32 *)h)[9];
            if (nseeds > O.S.scap) {
                if (attempt >= 2) { chip_set_error("seed scan: work list overflow"); return CATCHHIP_ENOMEM; }
                O.S.scap = nseeds;
                continue;
            }
            ctx->counters[1] = nseeds;
            for (int q = 0; q < 4; ++q) ctx->join_counters[q] = 0;
            ctx->seeds_dropped = (i64)nseeds - (i64)((volatile u32 *)h)[10];   // list entries without a seed
            P->seed_ratio_hint = std::max(P->seed_ratio_hint, (double)nseeds / (double)std::max<i64>(T->total, 1));   // see seed_capacity
            break;
        }
    } else {
        RawHits &H = O.H;
        H.want_seed = want_first;
        int rc = 0;
        if (P->nprobes > 0 && T->total > 0)
            rc = use_fast ? run_fast(ctx, P, T, mismatches, H, ts)
                          : run_general(ctx, P, T, mismatches, lcf_thres, island, H, ts);
        ts.stop();
        if (rc) return rc;
        TRY(bucket_prepare(O.B, nb, std::max(H.n, 1u), true));
        sink.rec = O.B.rec.p; sink.rank = O.B.rank.p; sink.bcnt = O.B.bcnt.p;
        HIP_TRY(hipMemsetAsync(O.B.bcnt.p, 0, sizeof(u32) * ((size_t)nb + 1), ctx->stream));
        HIP_TRY(hipMemsetAsync(O.B.res.p, 0, sizeof(u32) * 8, ctx->stream));
        O.nrec = H.n; O.nrec_dev = nullptr;
        tr.restart();
        if (H.n) {
            hipLaunchKernelGGL(hit_record_kernel, dim3((unsigned)div_up((i64)H.n, 256)), dim3(256), 0, ctx->stream,
                               (const u32 *)H.a.p, (const u32 *)H.b.p, H.has_end ? (const u32 *)H.c.p : (const u32 *)nullptr,
                               (u32)(P->L > 0 ? P->L : 0), H.n, (const u32 *)T->seq_off.p, (u32)T->nseq, sink);
            tr.launch();
        }
        TRY(bucket_finish_async(ctx, O.B, O.nrec, O.nrec_dev, true, !force_radix, tr, dedupe));
        tr.stop();
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ctx->h_pin, O.B.res.p, 8 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    ts.finish();
    const volatile u32 *h = (const volatile u32 *)ctx->h_pin;
    O.overflow = h[1] != 0 || force_radix;
    O.nhits = h[2]; O.maxbucket = h[3]; O.nrows = h[4]; O.lmax = h[5];
    ctx->counters[0] = O.nhits;
    tr.finish();
    return 0;
}

// gathers the facts of a deferred scan into the rows object (device side)
__global__ void rows_info_kernel(const u32 *__restrict__ res, const u32 *__restrict__ ctr, u32 scap,
                                 u32 *__restrict__ info) {
    const u32 t = threadIdx.x;
    if (t < 8) info[t] = res[t];
    else if (t < 12) info[t] = ctr[t - 8];
    else if (t == 12) info[t] = scap;
}

int chip_cover_scan_nosync(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T, i32 mismatches,
                           i32 lcf_thres, i32 island, i32 cover_extension, i32 mode, catchhip_rows **out) {
    *out = nullptr;
    if (P->nprobes == 0 || T->total == 0) return 1;
    if (!seed_path_ok(P, T, mismatches, lcf_thres, island)) return 1;
    if (!(mode == CATCHHIP_SCAN_SEED || (mode == CATCHHIP_SCAN_AUTO && !chip_test_env("CATCHHIP_SCAN_TILED")))) return 1;
    if (chip_test_env("CATCHHIP_ROWS_RADIX") || chip_test_env("CATCHHIP_SEED_CAP") || chip_test_env("CATCHHIP_FUSED_SYNC")) return 1;
    const i64 scap64 = seed_capacity(P, T);
    if (scap64 > ((i64)1 << 26)) return 1;   // keep the capacity-sized row arrays small
    HIP_TRY(hipSetDevice(ctx->device));
    PoolScope pool_scope(ctx);
    ScanOut O;
    O.S.scap = (u32)scap64;
    const u32 nb = (u32)P->nbuckets;
    HitSink sink;
    sink.bucket_of = P->bucket_identity ? nullptr : P->bucket_of.p;
    sink.seq_genome = T->seq_genome.p;
    sink.ext = (u32)cover_extension;
    if (P->has_groups != T->has_groups) return 1;   // the synchronous path reports the error
    if (P->has_groups) { sink.probe_group = P->group.p; sink.seq_group = T->seq_group.p; }
    TRY(bucket_prepare(O.B, nb, O.S.scap, false));
    sink.rec = O.B.rec.p; sink.rank = O.B.rank.p; sink.bcnt = O.B.bcnt.p;
    O.B.compact = true;
    sink.wcnt = O.B.wcnt.p;
    catchhip_rows *R = new catchhip_rows();
    R->ctx = ctx;
    R->total = T->total;
    R->ngenomes = T->ngenomes;
    R->h_genome_off = T->h_genome_off;
    R->grouped = P->has_groups && T->has_groups;
    R->deferred = true;
    R->n = O.S.scap;   // capacity; the row count is info[4]
    int rc = 0;
    do {
        if ((rc = R->genome_off.alloc((size_t)T->ngenomes + 1))) break;
        if ((rc = R->info.alloc(16))) break;
        if ((rc = R->set_id.alloc(R->n))) break;
        if ((rc = R->univ.alloc(R->n))) break;
        if ((rc = R->gs.alloc(R->n))) break;
        if ((rc = R->ge.alloc(R->n))) break;
        if (hipMemcpyAsync(R->genome_off.p, T->genome_off.p, sizeof(u32) * (T->ngenomes + 1),
                           hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { rc = CATCHHIP_EHIP; break; }
        PhaseTimer ts(ctx, PHASE_SCAN);
        if ((rc = run_seed_async(ctx, P, T, mismatches, O.S, sink, nb, O.B.res.p, ts))) break;
        ts.stop();
        PhaseTimer tr(ctx, PHASE_ROWS);
        if ((rc = bucket_finish_async(ctx, O.B, O.S.scap, O.S.ctr.p + 1, false, true, tr))) break;
        hipLaunchKernelGGL(rows_emit_kernel, dim3((unsigned)div_up(R->n, 256)), dim3(256), 0, ctx->stream,
                           (const u32 *)O.B.rstart.p, O.B.nb, (const u32 *)O.B.bstart.p,
                           P->bucket_identity ? (const i32 *)nullptr : (const i32 *)P->bucket_set.p,
                           (const uint4 *)O.B.S.p, (u32)R->n,
                           (const u32 *)(O.B.res.p + 4), R->set_id.p, R->univ.p, R->gs.p, R->ge.p,
                           (const u32 *)(O.B.res.p + 2), -1);
        hipLaunchKernelGGL(rows_info_kernel, dim3(1), dim3(64), 0, ctx->stream, (const u32 *)O.B.res.p,
                           (const u32 *)O.S.ctr.p, O.S.scap, R->info.p);
        tr.launch(2);
        tr.stop();
        if (hipGetLastError() != hipSuccess) { chip_set_error("cover scan: launch failed"); rc = CATCHHIP_EHIP; break; }
    } while (0);
    if (rc) { delete R; return rc; }
    // O's scratch goes back to this context's cache here; whatever reuses it is
    // ordered behind the kernels above on the context's stream
    *out = R;
    return 0;
}

// ------------------------------------------------------------------------
// first-discovery keys (catchhip_cover_scan_first_seen)
// ------------------------------------------------------------------------
// rows are sorted by (set id, global start); first row with (set, start) >= (s, x)
__device__ __forceinline__ u32 rows_lower_bound(const i32 *__restrict__ set_id, const u32 *__restrict__ gs, u32 n,
                                                i32 s, u32 x) {
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        const i32 ms = set_id[mid];
        if (ms < s || (ms == s && gs[mid] < x)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// One thread per accepted seed: the row holding the seed's k-mer, the first row
// of that row's (set, universe) group, and there the minimum of
// (k-mer position << 32 | caller's rank of the anchor entry).
__global__ void __launch_bounds__(256)
first_seen_kernel(int from_seeds, const uint4 *__restrict__ rec, const u32 *__restrict__ rank,
                  const u32 *__restrict__ spos, const u32 *__restrict__ sent, const u32 *__restrict__ nrec_dev, u32 cap,
                  const u32 *__restrict__ hp, const u32 *__restrict__ hd, const u32 *__restrict__ he,
                  const u32 *__restrict__ bucket_of, const i32 *__restrict__ bucket_set,
                  const u32 *__restrict__ anchor_order, const i32 *__restrict__ set_id, const i32 *__restrict__ univ,
                  const u32 *__restrict__ gs, const u32 *__restrict__ ge, const u32 *__restrict__ genome_off,
                  u32 nrows, unsigned long long *__restrict__ first) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 b, i, e;
    if (from_seeds) {
        if (t >= min(*nrec_dev, cap) || rank[t] == BK_NONE) return;
        b = rec[t].w; i = spos[t]; e = sent[t];
    } else {
        if (t >= cap) return;
        b = bucket_of[hp[t]]; i = hd[t]; e = he[t];
    }
    const i32 s = bucket_set[b];
    // last row with (set, start) <= (s, i): the cover range of an accepted seed contains its k-mer
    const u32 ub = rows_lower_bound(set_id, gs, nrows, s, i + 1u);
    if (ub == 0) return;
    const u32 r = ub - 1;
    if (set_id[r] != s || i >= ge[r]) return;   // cannot happen
    const u32 g0 = genome_off[univ[r]];
    const u32 head = rows_lower_bound(set_id, gs, nrows, s, g0);
    const unsigned long long key = ((unsigned long long)(i - g0) << 32) | (anchor_order ? anchor_order[e] : 0u);
    atomicMin(&first[head], key);
}

// every row takes the key of its group's first row
__global__ void __launch_bounds__(256)
first_seen_spread_kernel(const i32 *__restrict__ set_id, const i32 *__restrict__ univ, const u32 *__restrict__ gs,
                         const u32 *__restrict__ genome_off, u32 nrows, const unsigned long long *__restrict__ first,
                         unsigned long long *__restrict__ out) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const u32 head = rows_lower_bound(set_id, gs, nrows, set_id[r], genome_off[univ[r]]);
    out[r] = first[head];
}

static int cover_scan_impl(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T,
                           i32 mismatches, i32 lcf_thres, i32 island, i32 cover_extension, i32 mode, bool merge,
                           catchhip_rows **out, i64 *nrows, bool want_first = false,
                           const u32 *anchor_order = nullptr) {
    ARG_CHECK(ctx && P && T && out && cover_extension >= 0);
    PoolScope pool_scope(ctx);
    ARG_CHECK(P->ctx == ctx && T->ctx == ctx);
    *out = nullptr;
    if (nrows) *nrows = 0;
    HIP_TRY(hipSetDevice(ctx->device));
    const bool fast_ok = fast_path_ok(P, T, mismatches, lcf_thres, island);
    const bool seed_ok = seed_path_ok(P, T, mismatches, lcf_thres, island);
    if (mode == CATCHHIP_SCAN_FAST && !chip_test_env("CATCHHIP_TEST_HOOKS")) {
        // (round 6: the tiled O(P x G) scan is the tests' independent cross-check of the seed scans, not a product
        // path -- it answers only under CATCHHIP_TEST_HOOKS=1, and -DCATCHHIP_NO_TILED_SCAN builds a library without it)
        chip_set_error("cover_scan: CATCHHIP_SCAN_FAST (the tiled cross-check scan) is a test hook (CATCHHIP_TEST_HOOKS=1)");
        return CATCHHIP_EINVAL;
    }
#ifdef CATCHHIP_NO_TILED_SCAN
    if (mode == CATCHHIP_SCAN_FAST) { chip_set_error("cover_scan: this library was built without the tiled cross-check scan"); return CATCHHIP_EINVAL; }
#endif
    if (mode == CATCHHIP_SCAN_FAST && !fast_ok) {
        chip_set_error("cover_scan: fast-path preconditions do not hold");
        return CATCHHIP_EINVAL;
    }
    if (mode == CATCHHIP_SCAN_SEED && !seed_ok) {
        chip_set_error("cover_scan: seed-filter preconditions do not hold");
        return CATCHHIP_EINVAL;
    }
    // AUTO: all paths are exact.  With a full-length cover threshold on DNA the
    // seed scan (O(G + seeds)) is used, whatever the anchor table;
    // CATCHHIP_SCAN_FAST forces the tiled O(P*G) scan (pigeonhole anchors only),
    // CATCHHIP_SCAN_GENERAL the byte-exact seed join.
    catchhip_rows *R = new catchhip_rows();
    R->ctx = ctx;
    R->total = T->total;
    R->ngenomes = T->ngenomes;
    R->h_genome_off = T->h_genome_off;
    R->grouped = P->has_groups && T->has_groups;
    int rc = 0;
    do {
        if ((rc = R->genome_off.alloc((size_t)T->ngenomes + 1))) break;
        if (hipMemcpyAsync(R->genome_off.p, T->genome_off.p, sizeof(u32) * (T->ngenomes + 1),
                           hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { rc = CATCHHIP_EHIP; break; }
        if (P->nprobes == 0 || T->total == 0) break;   // no rows
        ScanOut O;
        if ((rc = scan_and_group(ctx, P, T, mismatches, lcf_thres, island, (u32)cover_extension, false, mode, O,
                                 !merge, want_first))) break;
        if (!merge && O.overflow) {
            chip_set_error("cover_ranges: a probe has more than %d cover ranges", BK_BIG);
            rc = CATCHHIP_EINVAL;
            break;
        }
        PhaseTimer tm(ctx, PHASE_ROWS, true);   // continues the row-build phase (adds to its time)
        if (!O.overflow) {
            R->n = O.nrows;
            R->lmax = O.lmax;
            if ((rc = R->set_id.alloc(R->n))) break;
            if ((rc = R->univ.alloc(R->n))) break;
            if ((rc = R->gs.alloc(R->n))) break;
            if ((rc = R->ge.alloc(R->n))) break;
            if (R->n) {
                hipLaunchKernelGGL(rows_emit_kernel, dim3((unsigned)div_up(R->n, 256)), dim3(256), 0, ctx->stream,
                                   (const u32 *)O.B.rstart.p, O.B.nb, (const u32 *)O.B.bstart.p,
                                   P->bucket_identity ? (const i32 *)nullptr : (const i32 *)P->bucket_set.p, (const uint4 *)O.B.S.p, (u32)R->n, (const u32 *)nullptr, R->set_id.p, R->univ.p,
                                   R->gs.p, R->ge.p, (const u32 *)nullptr, O.nhits == O.nrows ? 1 : 0);
                tm.launch();
                if (merge && O.B.bsum.p && O.B.bsum.n >= (size_t)O.B.nb && P->max_set_id < ((i64)1 << 31)) {
                    // the sets' total row lengths: what the first round of a full-coverage solve would count
                    const u32 ng = (u32)std::max<i64>(P->max_set_id + 1, (i64)(P->bucket_identity ? O.B.nb : 0));
                    if ((rc = R->gain0.alloc(ng))) break;
                    if (hipMemsetAsync(R->gain0.p, 0, sizeof(u32) * (size_t)ng, ctx->stream) != hipSuccess) { rc = CATCHHIP_EHIP; break; }
                    hipLaunchKernelGGL(rows_gain0_kernel, dim3((unsigned)div_up((i64)O.B.nb, 256)), dim3(256), 0, ctx->stream,
                                       (const unsigned long long *)O.B.bsum.p, O.B.nb,
                                       P->bucket_identity ? (const i32 *)nullptr : (const i32 *)P->bucket_set.p, ng, R->gain0.p);
                    R->gain0_n = ng;
                    tm.launch();
                }
            }
        } else {
            MergedRows M;
            if ((rc = build_rows_radix(ctx, O.B, O.nrec, O.nrec_dev, O.nhits, P->bucket_set.p, T->genome_off.p,
                                       (u32)T->ngenomes, P->max_set_id, M, tm, O.from_join ? &O.J : nullptr))) break;
            R->n = M.nmerged;
            if ((rc = R->set_id.alloc(R->n))) break;
            if ((rc = R->univ.alloc(R->n))) break;
            if ((rc = R->gs.alloc(R->n))) break;
            if ((rc = R->ge.alloc(R->n))) break;
            if (M.n) {
                DevBuf<u32> d_lmax;
                if ((rc = d_lmax.alloc(1))) break;
                if (hipMemsetAsync(d_lmax.p, 0, sizeof(u32), ctx->stream) != hipSuccess) { rc = CATCHHIP_EHIP; break; }
                hipLaunchKernelGGL(rows_compact_kernel, dim3((unsigned)div_up(M.n, 256)), dim3(256), 0, ctx->stream,
                                   M.keys.p, M.head.p, M.mend.p, M.seg.p, M.idx.p, M.n, R->set_id.p, R->univ.p,
                                   R->gs.p, R->ge.p, d_lmax.p);
                tm.launch();
                if (hipMemcpyAsync(ctx->h_pin, d_lmax.p, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                    hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = CATCHHIP_EHIP; break; }
                R->lmax = *(volatile u32 *)ctx->h_pin;
            }
        }
        if (want_first && R->n) {
            DevBuf<unsigned long long> first;
            DevBuf<u32> d_order;
            if ((rc = first.alloc(R->n)) || (rc = R->first_key.alloc(R->n))) break;
            if (anchor_order) {
                if ((rc = d_order.alloc((size_t)P->nent))) break;
                if (hipMemcpyAsync(d_order.p, anchor_order, sizeof(u32) * (size_t)P->nent, hipMemcpyHostToDevice,
                                   ctx->stream) != hipSuccess) { rc = CATCHHIP_EHIP; break; }
            }
            if (hipMemsetAsync(first.p, 0xff, sizeof(unsigned long long) * (size_t)R->n, ctx->stream) != hipSuccess) {
                rc = CATCHHIP_EHIP;
                break;
            }
            const u32 nsrc = O.from_seeds ? O.nrec : O.H.n;
            if (nsrc)
                hipLaunchKernelGGL(first_seen_kernel, dim3((unsigned)div_up((i64)nsrc, 256)), dim3(256), 0, ctx->stream,
                                   O.from_seeds ? 1 : 0, (const uint4 *)O.B.rec.p, (const u32 *)O.B.rank.p,
                                   (const u32 *)O.S.spos.p, (const u32 *)O.S.sent.p, O.nrec_dev, nsrc,
                                   (const u32 *)O.H.a.p, (const u32 *)O.H.d.p, (const u32 *)O.H.e.p,
                                   (const u32 *)P->bucket_of.p, (const i32 *)P->bucket_set.p,
                                   anchor_order ? (const u32 *)d_order.p : (const u32 *)nullptr, (const i32 *)R->set_id.p,
                                   (const i32 *)R->univ.p, (const u32 *)R->gs.p, (const u32 *)R->ge.p,
                                   (const u32 *)R->genome_off.p, (u32)R->n, first.p);
            hipLaunchKernelGGL(first_seen_spread_kernel, dim3((unsigned)div_up(R->n, 256)), dim3(256), 0, ctx->stream,
                               (const i32 *)R->set_id.p, (const i32 *)R->univ.p, (const u32 *)R->gs.p,
                               (const u32 *)R->genome_off.p, (u32)R->n, (const unsigned long long *)first.p,
                               R->first_key.p);
            tm.launch(2);
            // `first` and the order table go back to the pool after the synchronisation below
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = CATCHHIP_EHIP; break; }
        }
        tm.stop();
        // the scratch buffers of the build are released when O goes out of scope
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
            chip_set_error("cover_scan: row build failed: %s", hipGetErrorString(hipGetLastError()));
            rc = CATCHHIP_EHIP;
            break;
        }
        tm.finish_add();
    } while (0);
    if (rc) { delete R; return rc; }
    *out = R;
    if (nrows) *nrows = R->n;
    return 0;
}

extern "C" int catchhip_cover_scan(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T,
                                   i32 mismatches, i32 lcf_thres, i32 island, i32 cover_extension, i32 mode,
                                   catchhip_rows **out, i64 *nrows) {
    return cover_scan_impl(ctx, P, T, mismatches, lcf_thres, island, cover_extension, mode, true, out, nrows);
}

extern "C" int catchhip_cover_scan_first_seen(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T,
                                              i32 mismatches, i32 lcf_thres, i32 island, i32 cover_extension,
                                              i32 mode, const u32 *anchor_order, catchhip_rows **out, i64 *nrows) {
    ARG_CHECK(P);
    if (!P->sorted_unique) {
        chip_set_error("cover_scan_first_seen: the probes' anchors must have been given sorted by (probe, position) "
                       "without duplicates");
        return CATCHHIP_EINVAL;
    }
    return cover_scan_impl(ctx, P, T, mismatches, lcf_thres, island, cover_extension, mode, true, out, nrows, true,
                           anchor_order);
}

extern "C" int catchhip_rows_fetch_first_seen(catchhip_ctx *ctx, const catchhip_rows *R, u64 *first_key) {
    ARG_CHECK(ctx && R);
    if (R->n == 0) return 0;
    ARG_CHECK(first_key);
    if (!R->first_key.p) {
        chip_set_error("rows_fetch_first_seen: these rows do not come from catchhip_cover_scan_first_seen");
        return CATCHHIP_EINVAL;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(first_key, R->first_key.p, sizeof(u64) * R->n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int catchhip_cover_ranges(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T,
                                     i32 mismatches, i32 lcf_thres, i32 island, i32 cover_extension, i32 mode,
                                     catchhip_rows **out, i64 *nrows) {
    return cover_scan_impl(ctx, P, T, mismatches, lcf_thres, island, cover_extension, mode, false, out, nrows);
}

extern "C" int catchhip_tolerant_bp(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T,
                                    i32 mismatches, i32 lcf_thres, i32 island, i64 *bp_out) {
    ARG_CHECK(ctx && P && T && bp_out);
    PoolScope pool_scope(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    if (P->nprobes == 0 || T->total == 0) return 0;
    // merge per (probe, sequence): probe.find_probe_covers_in_sequence merges per sequence
    ScanOut O;
    TRY(scan_and_group(ctx, P, T, mismatches, lcf_thres, island, 0u, true, CATCHHIP_SCAN_AUTO, O));
    if (O.nhits == 0) return 0;
    std::vector<unsigned long long> h((size_t)P->nprobes);
    if (!O.overflow) {
        HIP_TRY(hipMemcpyAsync(h.data(), O.B.bsum.p, sizeof(unsigned long long) * P->nprobes, hipMemcpyDeviceToHost,
                               ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    } else {
        MergedRows M;
        PhaseTimer tm(ctx, PHASE_ROWS, true);
        TRY(build_rows_radix(ctx, O.B, O.nrec, O.nrec_dev, O.nhits, nullptr, T->seq_off.p, (u32)T->nseq, P->nprobes,
                             M, tm, O.from_join ? &O.J : nullptr));
        DevBuf<unsigned long long> bp;
        TRY(bp.alloc((size_t)P->nprobes));
        HIP_TRY(hipMemsetAsync(bp.p, 0, sizeof(unsigned long long) * P->nprobes, ctx->stream));
        hipLaunchKernelGGL(rows_bp_kernel, dim3((unsigned)div_up(M.n, 256)), dim3(256), 0, ctx->stream, M.keys.p,
                           M.head.p, M.mend.p, M.n, bp.p);
        tm.launch();
        tm.stop();
        HIP_TRY(hipMemcpyAsync(h.data(), bp.p, sizeof(unsigned long long) * P->nprobes, hipMemcpyDeviceToHost,
                               ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        tm.finish_add();
    }
    for (i64 i = 0; i < P->nprobes; ++i) bp_out[i] += (i64)h[i];
    return 0;
}

extern "C" int catchhip_rows_fetch(catchhip_ctx *ctx, const catchhip_rows *R, i32 *set_id, i32 *universe,
                                   i64 *start, i64 *end) {
    ARG_CHECK(ctx && R);
    if (R->n == 0) return 0;
    ARG_CHECK(set_id && universe && start && end);
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<u32> gs((size_t)R->n), ge((size_t)R->n);
    HIP_TRY(hipMemcpyAsync(set_id, R->set_id.p, sizeof(i32) * R->n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(universe, R->univ.p, sizeof(i32) * R->n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(gs.data(), R->gs.p, sizeof(u32) * R->n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ge.data(), R->ge.p, sizeof(u32) * R->n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (i64 i = 0; i < R->n; ++i) {
        i64 base = R->h_genome_off[universe[i]];
        start[i] = (i64)gs[i] - base;
        end[i] = (i64)ge[i] - base;
    }
    return 0;
}

// ------------------------------------------------------------------------
// per-universe / per-set statistics of a row table (coverage analysis at scale:
// the rows never leave the device)
// ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rows_stats_kernel(const i32 *__restrict__ set_id, const i32 *__restrict__ univ, const u32 *__restrict__ gs,
                  const u32 *__restrict__ ge, u32 n, unsigned long long *__restrict__ bm,
                  unsigned long long *__restrict__ total_len, unsigned long long *__restrict__ per_set) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const u32 s = gs[r], e = ge[r];
    atomicAdd(&total_len[univ[r]], (unsigned long long)(e - s));
    // rows are sorted by (set, start) and universes are contiguous: the first row of a
    // (set, universe) group counts one universe for its set
    if (per_set && (r == 0 || set_id[r - 1] != set_id[r] || univ[r - 1] != univ[r]))
        atomicAdd(&per_set[set_id[r]], 1ull);
    if (e > s) {
        const u32 w0 = s >> 6, w1 = (e - 1) >> 6;
        for (u32 w = w0; w <= w1; ++w) {
            u64 m = ~0ull;
            if (w == w0) m &= ~0ull << (s & 63);
            if (w == w1) m &= ~0ull >> (63 - ((e - 1) & 63));
            if ((bm[w] & m) != m) atomicOr(&bm[w], m);
        }
    }
}

__global__ void __launch_bounds__(256)
rows_union_kernel(const unsigned long long *__restrict__ bm, const u32 *__restrict__ genome_off,
                  unsigned long long *__restrict__ union_len) {
    __shared__ u32 part[4];
    const u32 u = blockIdx.x;
    const u32 s = genome_off[u], e = genome_off[u + 1];
    u32 c = 0;
    if (e > s) {
        const u32 w0 = s >> 6, w1 = (e - 1) >> 6;
        for (u32 w = w0 + threadIdx.x; w <= w1; w += blockDim.x) {
            u64 m = ~0ull;
            if (w == w0) m &= ~0ull << (s & 63);
            if (w == w1) m &= ~0ull >> (63 - ((e - 1) & 63));
            c += (u32)__popcll(bm[w] & m);
        }
    }
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, WAVE);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) union_len[u] = (unsigned long long)part[0] + part[1] + part[2] + part[3];
}

extern "C" int catchhip_rows_stats(catchhip_ctx *ctx, const catchhip_rows *R, i64 *total_len, i64 *union_len,
                                   i64 num_sets, i64 *universes_per_set) {
    ARG_CHECK(ctx && R && R->ctx == ctx && !R->deferred && total_len && union_len);
    ARG_CHECK(num_sets >= 0 && (num_sets == 0 || universes_per_set));
    PoolScope pool_scope(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t ng = (size_t)R->ngenomes, nw = (size_t)(R->total >> 6) + 2;
    if (ng == 0) return 0;
    DevBuf<unsigned long long> bm, tl, ul, ps;
    TRY(bm.alloc(nw));
    TRY(tl.alloc(ng));
    TRY(ul.alloc(ng));
    TRY(ps.alloc((size_t)num_sets + 1));
    HIP_TRY(hipMemsetAsync(bm.p, 0, sizeof(unsigned long long) * nw, s));
    HIP_TRY(hipMemsetAsync(tl.p, 0, sizeof(unsigned long long) * ng, s));
    HIP_TRY(hipMemsetAsync(ps.p, 0, sizeof(unsigned long long) * ((size_t)num_sets + 1), s));
    PhaseTimer tm(ctx, PHASE_ROWS);
    if (R->n) {
        // every set id must index universes_per_set
        hipLaunchKernelGGL(rows_stats_kernel, dim3((unsigned)div_up(R->n, 256)), dim3(256), 0, s,
                           (const i32 *)R->set_id.p, (const i32 *)R->univ.p, (const u32 *)R->gs.p, (const u32 *)R->ge.p,
                           (u32)R->n, bm.p, tl.p, num_sets ? ps.p : (unsigned long long *)nullptr);
    }
    hipLaunchKernelGGL(rows_union_kernel, dim3((unsigned)ng), dim3(256), 0, s, (const unsigned long long *)bm.p,
                       (const u32 *)R->genome_off.p, ul.p);
    tm.launch(2);
    HIP_TRY(hipGetLastError());
    static_assert(sizeof(i64) == sizeof(unsigned long long), "64-bit counters");
    HIP_TRY(hipMemcpyAsync(total_len, tl.p, sizeof(i64) * ng, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(union_len, ul.p, sizeof(i64) * ng, hipMemcpyDeviceToHost, s));
    if (num_sets) HIP_TRY(hipMemcpyAsync(universes_per_set, ps.p, sizeof(i64) * (size_t)num_sets, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    tm.finish();
    return 0;
}

// ------------------------------------------------------------------------
// adapter votes (catch/filter/adapter_filter.py:191-361) on a row table with
// first-discovery keys, one universe per sequence
// ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
av_key1_kernel(const unsigned long long *__restrict__ first_key, u32 n, u64 *__restrict__ keys, u32 *__restrict__ vals) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) { keys[r] = first_key[r]; vals[r] = r; }
}

__global__ void __launch_bounds__(256)
av_key2_kernel(const i32 *__restrict__ univ, const u32 *__restrict__ ge, const u32 *__restrict__ vals, u32 n,
               u64 *__restrict__ keys) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const u32 r = vals[i]; keys[i] = ((u64)(u32)univ[r] << 32) | ge[r]; }
}

// first index whose key's universe is >= u (keys sorted by (universe, end))
__device__ __forceinline__ u32 av_lower(const u64 *__restrict__ keys, u32 n, u32 u) {
    u32 lo = 0, hi = n;
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if ((u32)(keys[mid] >> 32) < u) lo = mid + 1; else hi = mid; }
    return lo;
}

// interval.schedule per sequence (catch/utils/interval.py:319-358): rows by end
// (ties: first-discovery key), take a row when it starts at or after the last
// taken end; a taken row marks its (probe, sequence) group
__global__ void __launch_bounds__(64)
av_schedule_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ perm, u32 n, u32 nuniv,
                   const i32 *__restrict__ set_id, const i32 *__restrict__ univ, const u32 *__restrict__ gs,
                   const u32 *__restrict__ ge, const u32 *__restrict__ genome_off, u32 *__restrict__ group_chosen) {
    const u32 u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nuniv) return;
    const u32 lo = av_lower(keys, n, u), hi = av_lower(keys, n, u + 1);
    u32 last_end = 0;
    bool any = false;
    for (u32 i = lo; i < hi; ++i) {
        const u32 r = perm[i];
        const u32 s = gs[r];
        if (!any || s >= last_end) {
            any = true;
            last_end = ge[r];
            // first row of the (set, universe) group in the set-sorted table
            const u32 head = rows_lower_bound(set_id, gs, n, set_id[r], genome_off[univ[r]]);
            group_chosen[head] = 1u;
        }
    }
}

// group heads (one per (probe, sequence)) keyed by universe
__global__ void __launch_bounds__(256)
av_heads_kernel(const i32 *__restrict__ set_id, const i32 *__restrict__ univ, u32 n, u32 *__restrict__ flag) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n) flag[r] = (r < n && (r == 0 || set_id[r - 1] != set_id[r] || univ[r - 1] != univ[r])) ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
av_headkeys_kernel(const u32 *__restrict__ flag, const u32 *__restrict__ at, const i32 *__restrict__ univ, u32 n,
                   u64 *__restrict__ keys, u32 *__restrict__ vals) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && flag[r]) { keys[at[r]] = (u64)(u32)univ[r]; vals[at[r]] = r; }
}

// The running totals over the sequences, in order (:330-358): one workgroup;
// per sequence every voter's vote is added plain or swapped, whichever gives
// the larger sum of per-probe majorities (swapped only if strictly larger).
__global__ void __launch_bounds__(1024)
av_tally_kernel(const u64 *__restrict__ hkeys, const u32 *__restrict__ heads, u32 nheads, u32 nuniv,
                const i32 *__restrict__ set_id, const u32 *__restrict__ group_chosen, const i64 *__restrict__ mult,
                i64 *__restrict__ cum_a, i64 *__restrict__ cum_b) {
    __shared__ long long s_plain[16], s_swap[16];
    __shared__ int s_flip;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32 lo = 0;
    while (lo < nheads) {
        const u32 u = (u32)hkeys[lo];            // uniform
        u32 hi = lo;                            // end of this universe's voters
        {
            u32 a = lo, b = nheads;
            while (a < b) { const u32 mid = (a + b) >> 1; if ((u32)hkeys[mid] <= u) a = mid + 1; else b = mid; }
            hi = a;
        }
        long long plain = 0, swp = 0;
        for (u32 i = lo + tid; i < hi; i += 1024) {
            const u32 h = heads[i];
            const i32 p = set_id[h];
            const long long a = group_chosen[h] ? 1 : 0, b = 1 - a, ca = cum_a[p], cb = cum_b[p], w = mult[p];
            const long long base = ca > cb ? ca : cb;
            const long long m1 = (ca + a > cb + b ? ca + a : cb + b), m2 = (ca + b > cb + a ? ca + b : cb + a);
            plain += w * (m1 - base);
            swp += w * (m2 - base);
        }
        for (int d = 32; d > 0; d >>= 1) { plain += __shfl_down(plain, d, WAVE); swp += __shfl_down(swp, d, WAVE); }
        if (lane == 0) { s_plain[wave] = plain; s_swap[wave] = swp; }
        __syncthreads();
        if (tid == 0) {
            long long P = 0, S = 0;
            for (int w = 0; w < 16; ++w) { P += s_plain[w]; S += s_swap[w]; }
            s_flip = S > P ? 1 : 0;
        }
        __syncthreads();
        const int flip = s_flip;
        for (u32 i = lo + tid; i < hi; i += 1024) {
            const u32 h = heads[i];
            const i32 p = set_id[h];
            const long long a = group_chosen[h] ? 1 : 0, b = 1 - a;
            cum_a[p] += flip ? b : a;
            cum_b[p] += flip ? a : b;
        }
        __syncthreads();
        lo = hi;
    }
}

extern "C" int catchhip_adapter_votes(catchhip_ctx *ctx, const catchhip_rows *R, i64 num_sets, const i64 *multiplicity,
                                      i64 *votes_a, i64 *votes_b) {
    ARG_CHECK(ctx && R && R->ctx == ctx && !R->deferred && num_sets >= 0);
    if (num_sets == 0) return 0;
    ARG_CHECK(votes_a && votes_b && multiplicity);
    PoolScope pool_scope(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    memset(votes_a, 0, sizeof(i64) * (size_t)num_sets);
    memset(votes_b, 0, sizeof(i64) * (size_t)num_sets);
    if (R->n == 0) return 0;
    if (!R->first_key.p) {
        chip_set_error("adapter_votes: these rows do not come from catchhip_cover_scan_first_seen");
        return CATCHHIP_EINVAL;
    }
    const u32 n = (u32)R->n, nuniv = (u32)R->ngenomes;
    DevBuf<u64> keys, keys_alt, hkeys, hkeys_alt;
    DevBuf<u32> perm, perm_alt, chosen, flag, at, tmp, heads, heads_alt;
    DevBuf<i64> d_mult, cum_a, cum_b;
    TRY(keys.alloc(n));
    TRY(perm.alloc(n));
    TRY(chosen.alloc((size_t)n + 1));
    TRY(flag.alloc((size_t)n + 1));
    TRY(at.alloc((size_t)n + 1));
    TRY(d_mult.alloc((size_t)num_sets));
    TRY(cum_a.alloc((size_t)num_sets));
    TRY(cum_b.alloc((size_t)num_sets));
    HIP_TRY(hipMemcpyAsync(d_mult.p, multiplicity, sizeof(i64) * (size_t)num_sets, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(cum_a.p, 0, sizeof(i64) * (size_t)num_sets, s));
    HIP_TRY(hipMemsetAsync(cum_b.p, 0, sizeof(i64) * (size_t)num_sets, s));
    HIP_TRY(hipMemsetAsync(chosen.p, 0, sizeof(u32) * ((size_t)n + 1), s));
    PhaseTimer tm(ctx, PHASE_ROWS);
    const unsigned nb = (unsigned)div_up((i64)n, 256);
    // rows by (universe, end, first-discovery key): two stable sorts, least significant key first
    hipLaunchKernelGGL(av_key1_kernel, dim3(nb), dim3(256), 0, s, (const unsigned long long *)R->first_key.p, n, keys.p, perm.p);
    TRY(chip_radix_sort_pairs(ctx, keys, keys_alt, perm, perm_alt, n, 64));
    hipLaunchKernelGGL(av_key2_kernel, dim3(nb), dim3(256), 0, s, (const i32 *)R->univ.p, (const u32 *)R->ge.p,
                       (const u32 *)perm.p, n, keys.p);
    TRY(chip_radix_sort_pairs(ctx, keys, keys_alt, perm, perm_alt, n, 64));
    hipLaunchKernelGGL(av_schedule_kernel, dim3((unsigned)div_up((i64)nuniv, 64)), dim3(64), 0, s, (const u64 *)keys.p,
                       (const u32 *)perm.p, n, nuniv, (const i32 *)R->set_id.p, (const i32 *)R->univ.p,
                       (const u32 *)R->gs.p, (const u32 *)R->ge.p, (const u32 *)R->genome_off.p, chosen.p);
    // voters: one per (set, universe) group, grouped by universe
    hipLaunchKernelGGL(av_heads_kernel, dim3((unsigned)div_up((i64)n + 1, 256)), dim3(256), 0, s, (const i32 *)R->set_id.p,
                       (const i32 *)R->univ.p, n, flag.p);
    TRY(chip_exclusive_scan_u32(ctx, flag.p, at.p, (i64)n + 1, tmp));
    HIP_TRY(hipMemcpyAsync(ctx->h_pin, at.p + n, sizeof(u32), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const u32 nheads = *(volatile u32 *)ctx->h_pin;
    TRY(hkeys.alloc(nheads));
    TRY(heads.alloc(nheads));
    hipLaunchKernelGGL(av_headkeys_kernel, dim3(nb), dim3(256), 0, s, (const u32 *)flag.p, (const u32 *)at.p,
                       (const i32 *)R->univ.p, n, hkeys.p, heads.p);
    TRY(chip_radix_sort_pairs(ctx, hkeys, hkeys_alt, heads, heads_alt, nheads, 32));
    hipLaunchKernelGGL(av_tally_kernel, dim3(1), dim3(1024), 0, s, (const u64 *)hkeys.p, (const u32 *)heads.p, nheads,
                       nuniv, (const i32 *)R->set_id.p, (const u32 *)chosen.p, (const i64 *)d_mult.p, cum_a.p, cum_b.p);
    tm.launch(60);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(votes_a, cum_a.p, sizeof(i64) * (size_t)num_sets, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(votes_b, cum_b.p, sizeof(i64) * (size_t)num_sets, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    tm.finish();
    return 0;
}

extern "C" int catchhip_rows_from_host(catchhip_ctx *ctx, const i32 *set_id, const i32 *universe,
                                       const i64 *start, const i64 *end, i64 nrows, const i64 *genome_len,
                                       i32 ngenomes, catchhip_rows **out) {
    ARG_CHECK(ctx && out && nrows >= 0 && ngenomes >= 0 && (ngenomes == 0 || genome_len));
    PoolScope pool_scope(ctx);
    ARG_CHECK(nrows == 0 || (set_id && universe && start && end));
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    catchhip_rows *R = new catchhip_rows();
    R->ctx = ctx;
    R->ngenomes = ngenomes;
    R->n = nrows;
    R->h_genome_off.assign((size_t)ngenomes + 1, 0);
    for (i32 g = 0; g < ngenomes; ++g) {
        if (genome_len[g] < 0) { delete R; chip_set_error("rows_from_host: negative genome length"); return CATCHHIP_EINVAL; }
        R->h_genome_off[g + 1] = R->h_genome_off[g] + genome_len[g];
    }
    R->total = R->h_genome_off[ngenomes];
    if (R->total >= ((i64)1 << 32) - 4096) { delete R; chip_set_error("rows_from_host: coordinate space too large"); return CATCHHIP_EINVAL; }
    std::vector<u32> gs((size_t)nrows), ge((size_t)nrows), go((size_t)ngenomes + 1);
    for (i32 g = 0; g <= ngenomes; ++g) go[g] = (u32)R->h_genome_off[g];
    for (i64 i = 0; i < nrows; ++i) {
        i32 u = universe[i];
        bool ok = u >= 0 && u < ngenomes && start[i] >= 0 && end[i] > start[i] && end[i] <= genome_len[u] && set_id[i] >= 0;
        if (ok && i > 0) {
            if (set_id[i] < set_id[i - 1]) ok = false;
            else if (set_id[i] == set_id[i - 1]) {
                if (u < universe[i - 1]) ok = false;
                else if (u == universe[i - 1] && start[i] <= end[i - 1]) ok = false;  // must be disjoint, non-touching
            }
        }
        if (!ok) { delete R; chip_set_error("rows_from_host: row %lld is invalid or out of order", (long long)i); return CATCHHIP_EINVAL; }
        gs[i] = (u32)(R->h_genome_off[u] + start[i]);
        ge[i] = (u32)(R->h_genome_off[u] + end[i]);
        R->lmax = std::max(R->lmax, ge[i] - gs[i]);
    }
    int rc = 0;
    do {
        if ((rc = R->set_id.alloc(nrows))) break;
        if ((rc = R->univ.alloc(nrows))) break;
        if ((rc = R->gs.alloc(nrows))) break;
        if ((rc = R->ge.alloc(nrows))) break;
        if ((rc = R->genome_off.alloc((size_t)ngenomes + 1))) break;
        hipStream_t s = ctx->stream;
        if ((nrows && (hipMemcpyAsync(R->set_id.p, set_id, sizeof(i32) * nrows, hipMemcpyHostToDevice, s) != hipSuccess ||
                       hipMemcpyAsync(R->univ.p, universe, sizeof(i32) * nrows, hipMemcpyHostToDevice, s) != hipSuccess ||
                       hipMemcpyAsync(R->gs.p, gs.data(), sizeof(u32) * nrows, hipMemcpyHostToDevice, s) != hipSuccess ||
                       hipMemcpyAsync(R->ge.p, ge.data(), sizeof(u32) * nrows, hipMemcpyHostToDevice, s) != hipSuccess)) ||
            hipMemcpyAsync(R->genome_off.p, go.data(), sizeof(u32) * (ngenomes + 1), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            chip_set_error("rows_from_host: upload failed");
            rc = CATCHHIP_EHIP;
        }
    } while (0);
    if (rc) { delete R; return rc; }
    *out = R;
    return 0;
}
