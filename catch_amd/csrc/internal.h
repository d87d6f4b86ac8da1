// Internal declarations shared by the translation units of libcatchhip.so.
// gfx950 (MI355X, CDNA4) only: 64-lane wavefronts are assumed throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/catchhip.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

#define WAVE 64

void chip_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                         \
    do {                                                                      \
        hipError_t e__ = (expr);                                              \
        if (e__ != hipSuccess) {                                              \
            chip_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,      \
                           hipGetErrorString(e__));                           \
            return CATCHHIP_EHIP;                                             \
        }                                                                     \
    } while (0)

#define TRY(expr)                                                             \
    do {                                                                      \
        int r__ = (expr);                                                     \
        if (r__ != 0) return r__;                                             \
    } while (0)

#define ARG_CHECK(cond)                                                       \
    do {                                                                      \
        if (!(cond)) {                                                        \
            chip_set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, \
                           #cond);                                            \
            return CATCHHIP_EINVAL;                                           \
        }                                                                     \
    } while (0)

// Environment switches.  Four are part of the product and read with getenv(): CATCHHIP_TIMING (host-side wall
// times on stderr), CATCHHIP_RCCL_PATH, CATCHHIP_POOL_SOFT_LIMIT_GB, CATCHHIP_GATHER_THREADS (README.md lists them
// with the Python side's).  Everything else is a TEST HOOK -- it forces one of several exact code paths (the radix
// row build, the set-parallel solver on a large instance, striped tiles ...) so that tests/ and bench.py can
// compare them -- and is only honoured when CATCHHIP_TEST_HOOKS=1 is set (tests/conftest.py and bench.py do).
static inline const char *chip_test_env(const char *name) {
    static const bool on = [] { const char *e = getenv("CATCHHIP_TEST_HOOKS"); return e && atoi(e) != 0; }();
    return on ? getenv(name) : nullptr;
}

// Caching device allocator (core.hip): hipMalloc/hipFree cost tens of
// microseconds and hipFree synchronises the device, which would dominate the
// millisecond-scale calls of this library.  Blocks are rounded up to a size
// class and kept on per-device free lists until the process exits; a block is
// only reused by work enqueued later on the context's single stream, after
// the host has observed the previous user's results.
void *chip_pool_alloc(size_t bytes);
// The cache is partitioned by owner (a context): entry points open a PoolScope
// so that blocks are only ever reused on the stream they were used on.
const void *chip_pool_set_owner(const void *owner);
void chip_pool_release_owner(const void *owner);
struct PoolScope {
    const void *prev;
    explicit PoolScope(const void *owner) : prev(chip_pool_set_owner(owner)) {}
    ~PoolScope() { chip_pool_set_owner(prev); }
};
void chip_pool_free(void *p);

// Device buffer owning a pooled region (returned to the pool in the destructor).
template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) chip_pool_free((void *)p);
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count) {
        release();
        n = count;
        size_t bytes = (count ? count : 1) * sizeof(T);
        p = (T *)chip_pool_alloc(bytes);
        if (!p) return CATCHHIP_ENOMEM;
        return 0;
    }
    // grow (content not preserved)
    int reserve(size_t count) {
        if (count <= n && p) return 0;
        return alloc(count);
    }
    void swap(DevBuf &o) {
        T *tp = p; p = o.p; o.p = tp;
        size_t tn = n; n = o.n; o.n = tn;
    }
};

enum { PHASE_SCAN = 0, PHASE_ROWS = 1, PHASE_GREEDY = 2, PHASE_NDF = 3, PHASE_GREEDY_ROUNDS = 4, PHASE_VERIFY = 5,
       PHASE_CLAIM = 6, PHASE_VCOUNT = 7, NPHASE = 8 };
#define CHIP_EVX 16   // event pairs for per-launch timing inside a batch of solver rounds (PHASE_CLAIM)

struct catchhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2 * NPHASE] = {};
    hipEvent_t evx[2 * CHIP_EVX] = {};
    double phase_ms[NPHASE] = {};
    i64 phase_launches[NPHASE] = {};
    i64 counters[8] = {};
    i64 ndf_counters[4] = {};      // last Hamming near-duplicate filter: probes, tables, pairs compared, edges
    i64 solver_counters[4] = {};   // row-parallel solver: records streamed, rows counted again, bitmap words read, owner words looked at
    i64 seeds_dropped = 0;   // of counters[1]: work-list entries the seed look-up's anchor-pair filter left empty
    i64 join_counters[4] = {};     // last key-grouped join: hit positions, pairs verified, lane slots of wave-wide runs, tasks of cut runs
    // pinned staging word(s) for small device->host reads
    u64 *h_pin = nullptr;
    // larger pinned staging area (grown on demand) so that result read-backs are
    // truly asynchronous and one synchronisation collects all of them
    void *h_big = nullptr;
    size_t h_big_bytes = 0;
    // RCCL (optional)
    void *comm = nullptr;
    int nranks = 1, rank = 0;
    int num_cus = 256;
};

// pinned host scratch of at least `bytes` (contents not preserved on growth)
int chip_pinned_reserve(catchhip_ctx *ctx, size_t bytes);

// elapsed time of a phase whose start/stop events were recorded earlier
void chip_phase_collect(catchhip_ctx *ctx, int phase);
// scan + row build with no host synchronisation (seed path only); returns 1
// when the inputs do not qualify (the caller then uses catchhip_cover_scan)
int chip_cover_scan_nosync(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T, i32 mismatches,
                           i32 lcf_thres, i32 island, i32 cover_extension, i32 mode, catchhip_rows **out);
// frontier solver on deferred rows; *retry = 1 when the device found the
// deferred scan unusable (overflow, long rows): redo through the synchronous calls
int chip_greedy_deferred(catchhip_ctx *ctx, catchhip_rows *R, i64 num_sets, const i64 *ranks, i64 *out_ids,
                         i64 *n_out, int *retry);

struct catchhip_targets {
    catchhip_ctx *ctx = nullptr;
    i64 total = 0;    // total bases (concatenated)
    i64 nseq = 0;
    i32 ngenomes = 0;
    i64 min_seq_len = 0;
    bool dna5 = false;   // alphabet subset of {A,C,G,T,N}
    bool has_n = false;  // contains a symbol other than A,C,G,T
    DevBuf<u8> bytes;        // raw characters
    DevBuf<u32> seq_off;     // nseq+1, global offsets (u32: total < 2^32)
    DevBuf<i32> seq_genome;  // nseq
    bool has_groups = false;
    DevBuf<i32> seq_group;   // nseq: instance of the sequence's genome (catchhip_targets_set_groups)
    i32 ngroups_set = 0;     // 1 + the largest group number
    DevBuf<u32> genome_off;  // ngenomes+1 global offset of each genome's first base
    i64 nwords = 0;          // 32-base words per plane (+ padding)
    DevBuf<u32> planes;      // 3 planes, SoA: plane b at planes + b*nwords
    // the same bits word-interleaved, [word] = {plane0, plane1, plane2, 0}: a
    // window of the seed-verify kernel is ONE run of 16-byte words (nwords entries)
    DevBuf<uint4> tq;
    std::vector<i64> h_seq_off;
    std::vector<i32> h_seq_genome;
    std::vector<i64> h_genome_off;
};

struct catchhip_probes {
    catchhip_ctx *ctx = nullptr;
    i64 nprobes = 0;
    i64 total = 0;
    i64 nent = 0;
    i32 k = 0;
    i32 L = 0;            // common probe length, or -1 if lengths differ
    bool dna5 = false;
    bool has_n = false;
    bool pigeonhole = false;  // anchors are exactly {0,k,2k,..,L-k} for every probe
    bool bucket_identity = false;   // bucket_of[p] == p and bucket_set[b] == b (probes from the device front end): no look-ups
    bool sorted_unique = false;   // the caller's anchor entries were sorted by (probe, position), no duplicates
    bool has_groups = false;
    DevBuf<i32> group;       // nprobes: instance of each probe (catchhip_probes_set_groups)
    // seeds per target base seen by earlier seed scans with these probes (sizes the work list)
    mutable double seed_ratio_hint = 0.0;
    i64 max_set_id = 0;
    DevBuf<u8> bytes;
    DevBuf<u32> probe_off;   // nprobes+1
    DevBuf<i32> set_id;      // nprobes
    i64 nbuckets = 0;        // distinct set ids
    DevBuf<u32> bucket_of;   // nprobes: dense rank of the probe's set id (row-table order)
    DevBuf<i32> bucket_set;  // nbuckets: set id of each bucket, ascending
    DevBuf<i32> ent_probe, ent_pos;
    // the same anchors sorted by (probe, position) + first entry of every probe (seed scan)
    DevBuf<u32> sent_probe, sent_pos, ent_ptr;
    i32 pwords = 0;          // 32-base words per probe (ceil(L/32))
    DevBuf<u32> planes;      // [probe][word][4] = planes 0,1,2 + pad per 32-base word
    DevBuf<uint2> w0;        // [probe] word 0 of planes 0/1 (the scan's 32-base filter), masked to L
};

// planes + word-0 image of equal-length DNA probes from p->bytes / p->probe_off (core.hip)
int chip_probes_pack_planes(catchhip_probes *p);

// near-duplicate filters on probes already on the device (ndf.hip): n rows of L
// characters / rows at probe_off[] (>= 16 bytes of slack after the last one);
// keep[] is a host array; d_keep_flags (instead, or nullptr) takes the verdicts as 0/1 words on the device
int chip_ndf_hamming_device(catchhip_ctx *ctx, const u8 *d_rows, i64 n, i32 L, const i32 *positions, i32 ntables,
                            i32 k, i32 dist_thres, u8 *keep, const u32 *d_grp, i64 ngroups, u32 *d_keep_flags = nullptr);
int chip_ndf_minhash_rows(catchhip_ctx *ctx, const u8 *d_rows, i64 n, i64 L, const u32 *d_grp, i64 ngroups, i32 kmer_size,
                          const i64 *ab, i32 ntables, i32 k, double dist_thres, u32 *d_keep_flags);
int chip_ndf_minhash_device(catchhip_ctx *ctx, const u8 *d_rows, const i64 *probe_off, i64 n, const i64 *group_off,
                            i64 ngroups, i32 kmer_size, const i64 *ab, i32 ntables, i32 k, double dist_thres,
                            u8 *keep);

// rows: cover intervals in GLOBAL coordinates of a targets object
struct catchhip_rows {
    catchhip_ctx *ctx = nullptr;
    i64 n = 0;
    i64 total = 0;       // size of the global coordinate space
    i32 ngenomes = 0;
    u32 lmax = 0;        // longest row (bases)
    // Deferred rows (fused scan + solve, never handed to the caller): the scan
    // has not been synchronised, n is only the capacity of the arrays and the
    // facts live on the device: info[0..7] = row-build results (scan.hip "res
    // words": [1] overflow, [2] hits, [4] rows, [5] longest row), info[8..11] =
    // seed counters ([9] seeds), info[12] = seed capacity
    bool deferred = false;
    // the scan's probes / targets carried group numbers (a union of independent instances, catchhip_*_set_groups): its
    // coordinate space is a row of unlike groups, which the row-parallel solver cuts into more, smaller tiles
    bool grouped = false;
    DevBuf<u32> info;
    mutable double seed_ratio_seen = 0.0;   // filled by the deferred solve: seeds per target base of the scan
    DevBuf<i32> set_id;
    DevBuf<i32> univ;
    DevBuf<u32> gs, ge;
    // catchhip_cover_scan_first_seen only: per row, the first-discovery key of its (set, universe) group
    DevBuf<unsigned long long> first_key;
    DevBuf<u32> genome_off;  // ngenomes+1
    std::vector<i64> h_genome_off;
    // gain0[s] = total length of set s's rows (the bucketed row build sums a bucket's merged rows on the way):
    // the first round's gains of a full-coverage solve, which then needs no count launch of its own; gain0_n = 0: not there
    DevBuf<u32> gain0;
    u32 gain0_n = 0;
};

// ---- timing helpers -----------------------------------------------------
struct PhaseTimer {
    catchhip_ctx *c;
    int phase;
    bool stopped = false;
    // keep = true continues a phase that was already timed once in this call:
    // finish_add() then adds to its time instead of replacing it
    PhaseTimer(catchhip_ctx *ctx, int ph, bool keep = false) : c(ctx), phase(ph) {
        if (keep) (void)hipEventRecord(c->ev[2 * phase], c->stream);
        else restart();
    }
    void restart() {
        stopped = false;
        c->phase_launches[phase] = 0;
        c->phase_ms[phase] = 0.0;
        (void)hipEventRecord(c->ev[2 * phase], c->stream);
    }
    void launch(i64 k = 1) { c->phase_launches[phase] += k; }
    // records the stop event (first call wins); finish() reads it after a sync
    void stop() {
        if (stopped) return;
        stopped = true;
        (void)hipEventRecord(c->ev[2 * phase + 1], c->stream);
    }
    void finish() {
        float ms = 0.f;
        stop();
        (void)hipEventSynchronize(c->ev[2 * phase + 1]);
        if (hipEventElapsedTime(&ms, c->ev[2 * phase], c->ev[2 * phase + 1]) == hipSuccess)
            c->phase_ms[phase] = ms;
    }
    void finish_add() {
        const double before = c->phase_ms[phase];
        finish();
        c->phase_ms[phase] += before;
    }
};

// ---- device primitives (primitives.hip) ---------------------------------
// exclusive prefix sum of n u32 values (in place allowed: out may equal in);
// if total != nullptr, *total (device u64) receives the grand total.
int chip_exclusive_scan_u32(catchhip_ctx *ctx, const u32 *in, u32 *out, i64 n,
                            DevBuf<u32> &tmp);
// Stable LSD radix sort of (u64 key, u32 value) pairs on bits [first_bit, first_bit + key_bits).
// Result ends in keys/vals (the alt buffers are scratch of the same size).
int chip_radix_sort_pairs(catchhip_ctx *ctx, DevBuf<u64> &keys, DevBuf<u64> &keys_alt,
                          DevBuf<u32> &vals, DevBuf<u32> &vals_alt, i64 n, int key_bits, int first_bit = 0);
// nseg segments of n keys each, side by side, every segment sorted on its own (one set of launches for all)
int chip_radix_sort_pairs_segments(catchhip_ctx *ctx, DevBuf<u64> &keys, DevBuf<u64> &keys_alt, DevBuf<u32> &vals,
                                   DevBuf<u32> &vals_alt, i64 n, i64 nseg, int key_bits, int first_bit = 0);

// v mod (2^31 - 1) for v < 2^63 without a 64-bit division: 2^31 = 1 (mod p),
// so the 31-bit digits of v add up (two folds leave at most p + 1)
__host__ __device__ static inline u32 mod_mersenne31(u64 v) {
    const u64 P = 0x7fffffffull;
    u64 t = (v & P) + (v >> 31);      // < 2^33
    t = (t & P) + (t >> 31);          // <= p + 3
    return (u32)(t >= P ? t - P : t);
}

// CPython <= 3.10 hash(str) of `len` ASCII characters under PYTHONHASHSEED=0: SipHash-2-4 with an all-zero
// key (PEP 456, Python/pyhash.c).  The reference hashes Probe objects by hash(seq_str) (catch/probe.py:
// 324-329): its sets of probes iterate in an order that follows from these values.
__host__ __device__ static inline long long chip_pyhash_seed0(const u8 *src, int len) {
    u64 v0 = 0x736f6d6570736575ull, v1 = 0x646f72616e646f6dull, v2 = 0x6c7967656e657261ull, v3 = 0x7465646279746573ull;
#define CHIP_SIPROUND do { \
    v0 += v1; v1 = (v1 << 13) | (v1 >> 51); v1 ^= v0; v0 = (v0 << 32) | (v0 >> 32); \
    v2 += v3; v3 = (v3 << 16) | (v3 >> 48); v3 ^= v2; \
    v0 += v3; v3 = (v3 << 21) | (v3 >> 43); v3 ^= v0; \
    v2 += v1; v1 = (v1 << 17) | (v1 >> 47); v1 ^= v2; v2 = (v2 << 32) | (v2 >> 32); } while (0)
    u64 b = (u64)len << 56;
    int n = len;
    const u8 *p = src;
    while (n >= 8) {
        u64 mi = 0;
        for (int i = 0; i < 8; ++i) mi |= (u64)p[i] << (8 * i);
        v3 ^= mi; CHIP_SIPROUND; CHIP_SIPROUND; v0 ^= mi;
        p += 8; n -= 8;
    }
    u64 t = 0;
    for (int i = 0; i < n; ++i) t |= (u64)p[i] << (8 * i);
    b |= t;
    v3 ^= b; CHIP_SIPROUND; CHIP_SIPROUND; v0 ^= b;
    v2 ^= 0xff; CHIP_SIPROUND; CHIP_SIPROUND; CHIP_SIPROUND; CHIP_SIPROUND;
#undef CHIP_SIPROUND
    long long x = (long long)((v0 ^ v1) ^ (v2 ^ v3));
    if (x == -1) x = -2;
    return x;
}
// order[] = the indices 0..n-1 in the order a CPython set iterates n distinct keys with these hashes
// after they were added in index order (core.hip)
void chip_pyset_order(const i64 *hash, i64 n, i64 *order);

static inline int ceil_log2_u64(u64 x) {
    int b = 0;
    while (b < 64 && ((u64)1 << b) < x) ++b;
    return b;
}
static inline i64 div_up(i64 a, i64 b) { return (a + b - 1) / b; }
