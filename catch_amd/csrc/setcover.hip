// K2: greedy multi-universe partial set cover on the device
// (catch/utils/set_cover.py:147-615 with use_intervalsets=True and cost == 1).
//
// State in HBM: one bit per base of the group's concatenated genomes ("still
// uncovered and part of the universe"), the cover rows as CSR (set -> (set,
// universe) segments -> rows), per-row / per-segment intersection counts and a
// per-set gain.  With cost == 1 the reference's ratio 1.0/gain orders sets
// exactly like the integer gain
//   gain(s) = sum_u min(left[u], |s_u ∩ U_u|)
// (1.0/n is injective for n < 2^53), ties go to the smallest set id (the
// iteration order of the reference's set of dense ids), ranks gate which sets
// may be considered (set_cover.py:497-526).
//
// Three solvers, all exact (catchhip_setcover_greedy picks one):
//
// Frontier rounds (setcover_batched.inc; every universe fully covered -- the
// default): all locally-maximal sets are accepted per
// round, two grid-wide launches each; the picks come back in the sequential
// order.
//
// Sequential solver (partial coverage, long rows): after grid-wide set-up
// kernels ONE persistent workgroup runs the whole greedy loop (a workgroup
// barrier costs a fraction of a microsecond where a kernel boundary costs
// several).  Per pick it (1) takes the arg-max of a two-level max
// structure over the gains, (2) clears the winner's bits, (3) re-counts only
// the rows that overlap the cleared ranges (found through a position-sorted
// row index) and patches the affected gains, (4) for universes whose
// remaining-need became binding re-evaluates min(left, count) for their
// segments, (5) refreshes the dirty blocks of the max structure.
//
// Multi-GPU solver (catchhip_comm_init): rank r evaluates the sets
// s % nranks == r with a full-recompute gain kernel, the winner is agreed by
// one RCCL all-reduce(MAX) of the packed 64-bit (gain, ~id) key per pick, and
// every rank applies it to its replica of the bitmap.
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <string>

#include "internal.h"

struct GreedyState {
    unsigned long long best_key;
    u32 n_need;    // universes with left > 0
    u32 cur_rank;  // dense rank index under consideration
    u32 nrank;
    u32 npicks;
    u32 done;      // 0 running, 1 finished, 2 rank list exhausted
    u32 iters;
    u32 lmax;      // longest row
    u32 smax;      // largest (set, universe) element count
    u32 fr_claim[2];   // frontier solver: some set of the current rank claimed, by round parity
    u32 fr_live[2];    // frontier solver: sizes of the live-set lists (large instances), by round parity
    unsigned long long prof[8];  // shader-clock ticks per phase (thread 0)
    unsigned long long n_wrows, n_recount, n_words;  // work counters
    u32 need_live, ticket;   // row-parallel solver, partial coverage: universes still in need (being counted) / workgroups done
    u32 nwon;                // row-parallel solver, full coverage: npicks when this round began -- the sets accepted in it are picks[nwon .. npicks)
};

// packed key = (gain << 32) | (0xFFFFFFFF - set id): gains are < 2^32 (a group's
// coordinate space is < 2^32 bases), set ids are u32
#define ID_BITS 32
#define ID_MASK 0xFFFFFFFFu
#define GAIN_BLOCK 256  // sets per block of the two-level max structure

#define LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#ifdef CATCHHIP_PROFILE   // per-phase shader-clock accounting (make CXXFLAGS+=-DCATCHHIP_PROFILE)
#define PROF(i) do { if (tid == 0) { unsigned long long t_ = __builtin_readcyclecounter(); st->prof[i] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define PROF(i) do { (void)t_prev; } while (0)
#endif

__device__ __forceinline__ u32 range_popcount(const u64 *__restrict__ bm, u32 s, u32 e) {
    u32 w0 = s >> 6, w1 = (e - 1) >> 6;
    u64 m0 = ~0ull << (s & 63);
    u64 m1 = ~0ull >> (63 - ((e - 1) & 63));
    if (w0 == w1) return (u32)__popcll(bm[w0] & m0 & m1);
    u32 c = (u32)__popcll(bm[w0] & m0);
    for (u32 w = w0 + 1; w < w1; ++w) c += (u32)__popcll(bm[w]);
    return c + (u32)__popcll(bm[w1] & m1);
}
// same, reading through L2 (words may have been cleared by atomics this launch)
// The common case (rows of <= 5 words, i.e. <= 257 bases) issues all its loads
// before the first use, so the words arrive in ONE memory round trip; a plain
// loop would pay one dependent L2/MALL round trip (~1k cycles) per word.
#define RP_MAXW 5
__device__ __forceinline__ u32 range_popcount_l2(const unsigned long long *bm, u32 s, u32 e) {
    u32 w0 = s >> 6, w1 = (e - 1) >> 6;
    u64 m0 = ~0ull << (s & 63);
    u64 m1 = ~0ull >> (63 - ((e - 1) & 63));
    const u32 nw = w1 - w0 + 1;
    if (nw <= RP_MAXW) {
        u64 v[RP_MAXW];
#pragma unroll
        for (u32 j = 0; j < RP_MAXW; ++j) v[j] = (j < nw) ? LD(&bm[w0 + j]) : 0ull;
        u32 c = 0;
#pragma unroll
        for (u32 j = 0; j < RP_MAXW; ++j) {
            u64 m = ~0ull;
            if (j == 0) m &= m0;
            if (j == nw - 1) m &= m1;
            if (j < nw) c += (u32)__popcll(v[j] & m);
        }
        return c;
    }
    if (w0 == w1) return (u32)__popcll(LD(&bm[w0]) & m0 & m1);
    u32 c = (u32)__popcll(LD(&bm[w0]) & m0);
    for (u32 w = w0 + 1; w < w1; ++w) c += (u32)__popcll(LD(&bm[w]));
    return c + (u32)__popcll(LD(&bm[w1]) & m1);
}

// the RP_MAXW = 5 consecutive 64-bit words at p with two 16-byte loads and one
// 8-byte load instead of five 8-byte ones: a lane's words are contiguous, but
// lanes sit on different rows, so every wave-level load costs one cache-line
// look-up per lane whatever its width (the round kernels are bound by that
// rate).  p is 8-byte aligned; the arrays have >= 8 words of slack at the end.
struct __attribute__((aligned(8))) u64x2 { unsigned long long a, b; };
// Only the loads that hold a word flagged in fl are issued (the others read as
// 0): a lane that sits out a load costs no look-up, and a 200-base row spans
// four words nine times out of ten, so the third load is rarely needed.
__device__ __forceinline__ void load_words5(const unsigned long long *p, unsigned long long (&v)[RP_MAXW], u32 fl) {
    u64x2 q0 = {0ull, 0ull}, q1 = {0ull, 0ull};
    unsigned long long q2 = 0ull;
    if (fl & 0x03u) q0 = *(const u64x2 *)p;
    if (fl & 0x0cu) q1 = *(const u64x2 *)(p + 2);
    if (fl & 0x10u) q2 = p[4];
    v[0] = q0.a; v[1] = q0.b; v[2] = q1.a; v[3] = q1.b; v[4] = q2;
}

// ------------------------------------------------------------------------
// set-up kernels (grid-wide)
// ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
set_ptr_kernel(const i32 *__restrict__ row_set, u32 nrows, u32 nsets, u32 *__restrict__ set_ptr) {
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nsets) return;
    u32 lo = 0, hi = nrows;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if ((u32)row_set[mid] < s) lo = mid + 1; else hi = mid;
    }
    set_ptr[s] = lo;
}

__global__ void __launch_bounds__(256)
seg_flag_kernel(const i32 *__restrict__ row_set, const i32 *__restrict__ row_univ, const u32 *__restrict__ gs,
                const u32 *__restrict__ ge, u32 nrows, u32 *__restrict__ flag, GreedyState *st) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    flag[r] = (r == 0 || row_set[r] != row_set[r - 1] || row_univ[r] != row_univ[r - 1]) ? 1u : 0u;
    atomicMax(&st->lmax, ge[r] - gs[r]);
}

__global__ void __launch_bounds__(256)
seg_fill_kernel(const u32 *__restrict__ flag, const u32 *__restrict__ idx, const i32 *__restrict__ row_set,
                const i32 *__restrict__ row_univ, u32 nrows, u32 nseg, u32 *__restrict__ seg_row,
                u32 *__restrict__ seg_univ, u32 *__restrict__ seg_set, u32 *__restrict__ row_seg) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) seg_row[nseg] = nrows;
    if (r >= nrows) return;
    u32 q = idx[r] + flag[r] - 1;  // segment of this row
    row_seg[r] = q;
    if (!flag[r]) return;
    seg_row[q] = r;
    seg_univ[q] = (u32)row_univ[r];
    seg_set[q] = (u32)row_set[r];
}

__global__ void __launch_bounds__(256)
set_seg_ptr_kernel(const u32 *__restrict__ set_ptr, const u32 *__restrict__ idx, u32 nrows, u32 nsets,
                   u32 nseg, u32 *__restrict__ set_seg_ptr) {
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nsets) return;
    u32 r = set_ptr[s];
    set_seg_ptr[s] = (r < nrows) ? idx[r] : nseg;
}

// universe bitmap = union of every set's rows (set_cover.py:302-320)
__global__ void __launch_bounds__(256)
bitmap_build_kernel(const u32 *__restrict__ gs, const u32 *__restrict__ ge, u32 nrows,
                    unsigned long long *__restrict__ bm) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    u32 s = gs[r], e = ge[r];
    u32 w0 = s >> 6, w1 = (e - 1) >> 6;
    u64 m0 = ~0ull << (s & 63);
    u64 m1 = ~0ull >> (63 - ((e - 1) & 63));
    if (w0 == w1) { atomicOr(&bm[w0], m0 & m1); return; }
    atomicOr(&bm[w0], m0);
    for (u32 w = w0 + 1; w < w1; ++w) atomicOr(&bm[w], ~0ull);
    atomicOr(&bm[w1], m1);
}

// |U_u| per universe: one workgroup per universe
__global__ void __launch_bounds__(256)
universe_size_kernel(const u64 *__restrict__ bm, const u32 *__restrict__ genome_off, u32 nuniv,
                     u32 *__restrict__ usize) {
    __shared__ u32 part[4];
    u32 u = blockIdx.x;
    u32 s = genome_off[u], e = genome_off[u + 1];
    u32 c = 0;
    if (e > s) {
        u32 w0 = s >> 6, w1 = (e - 1) >> 6;
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
    if (threadIdx.x == 0) usize[u] = part[0] + part[1] + part[2] + part[3];
}

// num_that_can_be_uncovered / num_left_to_cover (set_cover.py:362-373):
// int(len(U) - p * len(U)) in IEEE double, no fused multiply-add.
__global__ void __launch_bounds__(256)
universe_need_kernel(const u32 *__restrict__ usize, const double *__restrict__ p, u32 nuniv,
                     u32 *__restrict__ can, u32 *__restrict__ left, GreedyState *__restrict__ st) {
    u32 u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nuniv) return;
    double n = (double)usize[u];
    double prod = __dmul_rn(p ? p[u] : 1.0, n);
    double diff = __dsub_rn(n, prod);
    long long c = (long long)diff;  // truncation toward zero == Python int()
    if (c < 0) c = 0;
    if (c > (long long)usize[u]) c = usize[u];
    can[u] = (u32)c;
    u32 l = usize[u] - (u32)c;
    left[u] = l;
    if (l > 0) atomicAdd(&st->n_need, 1u);
}

__global__ void greedy_start_kernel(GreedyState *st) {
    if (st->n_need == 0) st->done = 1;
}

// initial per-row counts (every row lies wholly inside the fresh universe)
__global__ void __launch_bounds__(256)
rowcnt_init_kernel(const u32 *__restrict__ gs, const u32 *__restrict__ ge, const u32 *__restrict__ row_seg,
                   u32 nrows, u32 *__restrict__ segcnt) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    atomicAdd(&segcnt[row_seg[r]], ge[r] - gs[r]);
}

__global__ void __launch_bounds__(256)
seg_init_kernel(const u32 *__restrict__ segcnt, const u32 *__restrict__ seg_univ,
                const u32 *__restrict__ seg_set, const u32 *__restrict__ left, u32 nseg,
                u32 *__restrict__ segcontrib, GreedyState *__restrict__ st,
                u64 *__restrict__ ukeys, u32 *__restrict__ uvals) {
    u32 q = blockIdx.x * blockDim.x + threadIdx.x;
    u32 c = 0;
    if (q < nseg) {
        u32 u = seg_univ[q], l = left[u];
        c = segcnt[q];
        segcontrib[q] = c < l ? c : l;
        ukeys[q] = u;   // for the per-universe segment index
        uvals[q] = q;
    }
    // largest segment count (bounds when min(left, count) can bind): one
    // atomic per wavefront instead of one contended atomic per segment
    for (int d = 32; d > 0; d >>= 1) { u32 o = __shfl_down(c, d, WAVE); c = o > c ? o : c; }
    if ((threadIdx.x & 63) == 0 && c) atomicMax(&st->smax, c);
}

// initial gain of every set = sum of its segments' contributions (no atomics)
__global__ void __launch_bounds__(256)
gain_init_kernel(const u32 *__restrict__ segcontrib, const u32 *__restrict__ set_seg_ptr, u32 nsets,
                 u32 *__restrict__ gain) {
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsets) return;
    u32 g = 0;
    for (u32 q = set_seg_ptr[s]; q < set_seg_ptr[s + 1]; ++q) g += segcontrib[q];
    gain[s] = g;
}

__global__ void __launch_bounds__(256)
pos_key_kernel(const u32 *__restrict__ gs, u32 nrows, u64 *__restrict__ keys, u32 *__restrict__ vals) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    keys[r] = gs[r];
    vals[r] = r;
}

// position-sorted row table {gs, ge, row, set} and its coarse bucket index
__global__ void __launch_bounds__(256)
pent_fill_kernel(const u32 *__restrict__ pos_row, const u32 *__restrict__ gs, const u32 *__restrict__ ge,
                 const i32 *__restrict__ row_set, const u32 *__restrict__ row_seg, u32 nrows,
                 uint4 *__restrict__ pent, u32 *__restrict__ prowcnt) {
    u32 y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= nrows) return;
    u32 r = pos_row[y];
    pent[y] = make_uint4(gs[r], ge[r], row_seg[r], (u32)row_set[r]);
    prowcnt[y] = ge[r] - gs[r];   // every row lies wholly inside the fresh universe
}

// rows in set order with what the apply / re-count phases need in one load
__global__ void __launch_bounds__(256)
wrow_fill_kernel(const u32 *__restrict__ gs, const u32 *__restrict__ ge, const i32 *__restrict__ row_set,
                 const i32 *__restrict__ row_univ, const u32 *__restrict__ row_seg,
                 const u32 *__restrict__ can, u32 nrows, uint4 *__restrict__ wrow) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    u32 prev = (r > 0 && row_set[r - 1] == row_set[r] && row_univ[r - 1] == row_univ[r]) ? ge[r - 1] : 0u;
    // bit 31 of the segment field: this universe may stay partly uncovered
    // (can > 0), so min(left, count) has to be evaluated exactly
    wrow[r] = make_uint4(gs[r], ge[r], prev, row_seg[r] | (can[row_univ[r]] ? 0x80000000u : 0u));
}

__global__ void __launch_bounds__(256)
bucket_kernel(const u64 *__restrict__ pos_key, u32 nrows, u32 nbuckets, int shift, u32 *__restrict__ bucket) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbuckets) return;
    u64 want = (u64)b << shift;
    u32 lo = 0, hi = nrows;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if (pos_key[mid] < want) lo = mid + 1; else hi = mid;
    }
    bucket[b] = lo;
}

__global__ void __launch_bounds__(256)
useg_ptr_kernel(const u64 *__restrict__ ukeys, u32 nseg, u32 nuniv, u32 *__restrict__ useg_ptr) {
    u32 u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u > nuniv) return;
    u32 lo = 0, hi = nseg;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if ((u32)ukeys[mid] < u) lo = mid + 1; else hi = mid;
    }
    useg_ptr[u] = lo;
}

// ------------------------------------------------------------------------
// persistent single-workgroup greedy loop
// ------------------------------------------------------------------------
struct GreedyArgs {
    unsigned long long *bm;
    const uint4 *wrow;    // rows in set order: {gs, ge, prev_ge (same universe, else 0), segment}
    const u32 *set_ptr, *set_seg_ptr;
    const u32 *seg_univ, *seg_set;
    const uint4 *pent;    // rows sorted by start: {gs, ge, segment, set}
    u32 *prowcnt;         // |row ∩ universe| per sorted slot
    const u32 *bucket;    // first sorted slot with gs >= (b << BUCKET_SHIFT)
    const u32 *useg_ptr, *useg;
    const u32 *can, *rank;
    u32 *usize, *left, *segcnt, *segcontrib, *gain;
    u32 *picked, *picks, *dirty;
    GreedyState *st;
    u32 nrows, nsets, nuniv, chunk;
};

#define GW_THREADS 1024
#define GW_WAVES (GW_THREADS / WAVE)
#define GW_MAXBIND 1024
#define GW_MAXWSEG 4096    // winner segments with LDS accumulators
#define GW_MAXDIRTY 4096   // dirty segments staged in LDS
#define GW_CAND 4          // candidate rows fetched per lane per step
#define BUCKET_SHIFT 5

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        unsigned long long o = __shfl_down(v, d, WAVE);
        v = o > v ? o : v;
    }
    return v;
}

// contribution min(left, count) of segment q changed? patch the set's gain.
__device__ __forceinline__ void patch_segment(const GreedyArgs &a, u32 q, u32 *s_cdirty) {
    const u32 ss = a.seg_set[q];
    if (LD(&a.picked[ss])) return;
    const u32 l = LD(&a.left[a.seg_univ[q]]), c = LD(&a.segcnt[q]);
    const u32 nc = c < l ? c : l;
    const u32 oc = atomicExch(&a.segcontrib[q], nc);
    if (oc != nc) {
        atomicAdd(&a.gain[ss], nc - oc);
        s_cdirty[ss / a.chunk] = 1u;
    }
}

__global__ void __launch_bounds__(GW_THREADS)
greedy_wg_kernel(GreedyArgs a) {
    __shared__ unsigned long long s_red[GW_WAVES];
    __shared__ u32 s_cdirty[GW_THREADS];  // this thread's chunk of sets must be re-scanned
    __shared__ u32 s_bind[GW_MAXBIND];
    __shared__ u32 s_clr[GW_MAXWSEG];     // bits cleared per winner segment
    __shared__ u32 s_dirty[GW_MAXDIRTY];
    __shared__ u32 s_nbind, s_ndirty, s_need, s_stop, s_npicks, s_iters;
    __shared__ unsigned long long s_nwrows, s_nrecount, s_nwords;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    GreedyState *st = a.st;

    if (tid == 0) {
        s_need = st->n_need; s_stop = st->done;
        s_nbind = 0; s_ndirty = 0; s_npicks = 0; s_iters = 0;
        s_nwrows = 0; s_nrecount = 0; s_nwords = 0;
    }
    s_cdirty[tid] = 1u;
    __syncthreads();
    if (s_stop) return;
    const u32 lmax = st->lmax, smax = st->smax;
    u32 cnt_recount = 0, cnt_words = 0;   // per-thread work counters
    const u32 c0 = tid * a.chunk;                                  // this thread's sets
    const u32 c1 = min(a.nsets, c0 + a.chunk);
    unsigned long long ckey = 0;                                   // best key in the chunk
    u32 my_rank = 0xffffffffu;                                     // rank the chunk scan is valid for
    u32 cur_rank = st->cur_rank, stop = 0;                         // uniform across the workgroup
    const u32 nrank = st->nrank;
    unsigned long long t_prev = __builtin_readcyclecounter();

    for (;;) {
        // ---- A: arg-max.  Only chunks whose gains changed are re-scanned ----
        if (s_cdirty[tid] || my_rank != cur_rank) {
            s_cdirty[tid] = 0u;
            my_rank = cur_rank;
            unsigned long long k = 0;
            for (u32 s0 = c0; s0 < c1; s0 += 16) {
                // gains and ranks of 16 sets in flight together: one round trip
                u32 g[16], rk[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const u32 sj = min(s0 + j, c1 - 1);
                    g[j] = LD(&a.gain[sj]);
                    rk[j] = a.rank[sj];
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    // picked sets hold gain 0 (see below); other ranks are masked here
                    if (s0 + j < c1 && g[j] && rk[j] == my_rank) {
                        unsigned long long kk = ((unsigned long long)g[j] << ID_BITS) |
                                                (unsigned long long)(ID_MASK - (s0 + j));
                        k = kk > k ? kk : k;
                    }
                }
            }
            ckey = k;
        }
        {
            unsigned long long k = wave_max_u64(ckey);
            if (lane == 0) s_red[wave] = k;
        }
        __syncthreads();
        // every wave folds the 16 partial maxima itself: no second barrier and no
        // global access between the scan and the apply phase
        unsigned long long key = wave_max_u64(lane < GW_WAVES ? s_red[lane] : 0ull);
        key = __shfl(key, 0, WAVE);
        PROF(0);
        if ((key >> ID_BITS) == 0) {
            // no set of this rank covers anything still needed: next rank
            // (set_cover.py:522-526); chunks re-scan because my_rank != cur_rank
            cur_rank++;
            if (tid == 0) s_iters++;
            if (cur_rank >= nrank) { stop = 2; break; }
            __syncthreads();   // s_red is rewritten by the next scan
            continue;
        }
        const u32 s = ID_MASK - (u32)(key & ID_MASK);
        const u32 r0 = a.set_ptr[s], nw = a.set_ptr[s + 1] - r0;
        const u32 q0 = a.set_seg_ptr[s], nq = a.set_seg_ptr[s + 1] - q0;
        for (u32 i = tid; i < nq && i < GW_MAXWSEG; i += GW_THREADS) s_clr[i] = 0u;
        if (tid == 0) {
            s_iters++;
            ST(&a.picked[s], 1u);
            ST(&a.gain[s], 0u);          // never eligible again; its rows are skipped below
            a.picks[s_npicks++] = s;
            s_cdirty[s / a.chunk] = 1u;
            s_nbind = 0;
            s_ndirty = 0;
            s_nwrows += nw;
        }
        __syncthreads();

        // ---- B: apply: remove the winner's elements (set_cover.py:531-550) --
        for (u32 i = tid; i < nw; i += GW_THREADS) {
            const uint4 wr = a.wrow[r0 + i];
            const u32 x = wr.x, e = wr.y;
            const u32 w0 = x >> 6, w1 = (e - 1) >> 6, nwd = w1 - w0 + 1;
            const u64 m0 = ~0ull << (x & 63), m1 = ~0ull >> (63 - ((e - 1) & 63));
            u32 cleared = 0;
            if (nwd <= RP_MAXW) {
                // all returning atomics in flight together: one round trip
                u64 old[RP_MAXW], msk[RP_MAXW];
#pragma unroll
                for (u32 j = 0; j < RP_MAXW; ++j) {
                    u64 m = ~0ull;
                    if (j == 0) m &= m0;
                    if (j == nwd - 1) m &= m1;
                    msk[j] = m;
                    old[j] = (j < nwd) ? atomicAnd(&a.bm[w0 + j], ~m) : 0ull;
                }
#pragma unroll
                for (u32 j = 0; j < RP_MAXW; ++j)
                    if (j < nwd) cleared += (u32)__popcll(old[j] & msk[j]);
            } else {
                for (u32 w = w0; w <= w1; ++w) {
                    u64 m = ~0ull;
                    if (w == w0) m &= m0;
                    if (w == w1) m &= m1;
                    u64 old = atomicAnd(&a.bm[w], ~m);
                    cleared += (u32)__popcll(old & m);
                }
            }
            if (cleared) {
                const u32 li = (wr.w & 0x7fffffffu) - q0;
                if (li < GW_MAXWSEG) atomicAdd(&s_clr[li], cleared);
                else atomicSub(&a.usize[a.seg_univ[wr.w & 0x7fffffffu]], cleared);
            }
        }
        __syncthreads();
        PROF(1);

        // ---- C1: remaining need per touched universe (single writer) ---------
        // (wave 0 only, so that its dependent loads overlap the other waves' C2 chain)
        if (wave == 0) for (u32 i = lane; i < nq; i += WAVE) {
            const u32 u = a.seg_univ[q0 + i];
            u32 n = LD(&a.usize[u]);
            if (i < GW_MAXWSEG && s_clr[i]) { n -= s_clr[i]; ST(&a.usize[u], n); }
            const u32 c = a.can[u];
            const u32 nl = n > c ? n - c : 0u;
            const u32 ol = LD(&a.left[u]);
            if (nl != ol) {
                ST(&a.left[u], nl);
                if (ol > 0 && nl == 0) atomicSub(&s_need, 1u);
                // min(left, count) can only bind when part of the universe may
                // stay uncovered and the need fell below the largest count
                if (c > 0 && nl < smax) {
                    u32 slot = atomicAdd(&s_nbind, 1u);
                    if (slot < GW_MAXBIND) s_bind[slot] = u;
                }
            }
        }
        // ---- C2: re-count the rows that overlap a cleared range --------------
        // one 16-lane group per winner row; candidates come from the
        // position-sorted row table through a coarse bucket index
        if (wave != 0) {
            const int sub = tid & 15;
            for (u32 i = (tid - WAVE) >> 4; i < nw; i += (GW_THREADS - WAVE) / 16) {
                const uint4 wr = a.wrow[r0 + i];
                const u32 x = wr.x, e = wr.y, prev = wr.z;
                const bool partial = (wr.w >> 31) != 0;   // min(left,count) may bind in this universe
                const u32 lo_start = x > lmax ? x - lmax : 0u;      // earlier rows cannot reach x
                u32 y = a.bucket[lo_start >> BUCKET_SHIFT] + sub;
                bool more = true;
                while (more) {
                    uint4 c[GW_CAND];
#pragma unroll
                    for (int j = 0; j < GW_CAND; ++j) {
                        const u32 yy = y + 16 * j;
                        c[j] = a.pent[min(yy, a.nrows - 1)];
                        if (yy >= a.nrows) c[j].x = c[j].y = 0xffffffffu;   // past the end
                    }
                    u32 nc[GW_CAND], oc[GW_CAND], pk[GW_CAND], cnw[GW_CAND];
                    u64 v[GW_CAND][RP_MAXW];
                    bool use[GW_CAND];
                    // every load below is unconditional (clamped to a valid
                    // address when unused) so that all candidates' words arrive
                    // in one memory round trip instead of one per candidate
#pragma unroll
                    for (int j = 0; j < GW_CAND; ++j) {
                        // overlaps the range and was not already visited from
                        // the previous range of this universe
                        use[j] = c[j].x < e && c[j].y > x && prev <= c[j].x;
                        const u32 cs = use[j] ? c[j].x : 0u, ce = use[j] ? c[j].y : 1u;
                        const u32 w0 = cs >> 6;
                        cnw[j] = ((ce - 1) >> 6) - w0 + 1;
#pragma unroll
                        for (u32 t = 0; t < RP_MAXW; ++t) v[j][t] = LD(&a.bm[w0 + min(t, cnw[j] - 1)]);
                        pk[j] = LD(&a.picked[use[j] ? c[j].w : 0u]);   // rows of chosen sets are dead
                        oc[j] = a.prowcnt[use[j] ? y + 16 * j : 0u];
                    }
#pragma unroll
                    for (int j = 0; j < GW_CAND; ++j) {
                        nc[j] = 0;
                        if (pk[j]) use[j] = false;
                        if (!use[j]) continue;
                        cnt_recount++;
                        cnt_words += cnw[j];
                        if (cnw[j] > RP_MAXW) { nc[j] = range_popcount_l2(a.bm, c[j].x, c[j].y); continue; }
                        const u64 m0 = ~0ull << (c[j].x & 63), m1 = ~0ull >> (63 - ((c[j].y - 1) & 63));
#pragma unroll
                        for (u32 t = 0; t < RP_MAXW; ++t) {
                            u64 m = ~0ull;
                            if (t == 0) m &= m0;
                            if (t == cnw[j] - 1) m &= m1;
                            if (t < cnw[j]) nc[j] += (u32)__popcll(v[j][t] & m);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < GW_CAND; ++j) {
                        if (use[j] && oc[j] != nc[j]) {
                            const u32 d = oc[j] - nc[j];
                            a.prowcnt[y + 16 * j] = nc[j];
                            if (partial) {
                                atomicSub(&a.segcnt[c[j].z], d);
                                const u32 slot = atomicAdd(&s_ndirty, 1u);
                                if (slot < GW_MAXDIRTY) s_dirty[slot] = c[j].z; else a.dirty[slot] = c[j].z;
                            } else {
                                // fully covered universes: contribution == count
                                atomicSub(&a.gain[c[j].w], d);
                                s_cdirty[c[j].w / a.chunk] = 1u;
                            }
                        }
                    }
                    more = c[GW_CAND - 1].x < e;   // sorted by start: stop once past the range
                    y += 16 * GW_CAND;
                }
            }
        }
        DRAIN();
        __syncthreads();
        PROF(2);

        // ---- D: universes with partial cover: exact min(left, count) ---------
        if (s_ndirty | s_nbind) {
            const u32 nd = s_ndirty;
            for (u32 i = tid; i < nd; i += GW_THREADS)
                patch_segment(a, i < GW_MAXDIRTY ? s_dirty[i] : LD(&a.dirty[i]), s_cdirty);
            const u32 nb = s_nbind;
            if (nb > GW_MAXBIND) {
                for (u32 u = wave; u < a.nuniv; u += GW_WAVES) {
                    if (a.can[u] == 0 || LD(&a.left[u]) >= smax) continue;
                    for (u32 z = a.useg_ptr[u] + lane; z < a.useg_ptr[u + 1]; z += WAVE)
                        patch_segment(a, a.useg[z], s_cdirty);
                }
            } else {
                for (u32 i = wave; i < nb; i += GW_WAVES) {
                    const u32 u = s_bind[i];
                    for (u32 z = a.useg_ptr[u] + lane; z < a.useg_ptr[u + 1]; z += WAVE)
                        patch_segment(a, a.useg[z], s_cdirty);
                }
            }
            DRAIN();
            __syncthreads();
        }
        PROF(3);
        if (s_need == 0) break;
    }
    atomicAdd(&s_nrecount, (unsigned long long)cnt_recount);
    atomicAdd(&s_nwords, (unsigned long long)cnt_words);
    __syncthreads();
    if (tid == 0) {
        st->done = stop == 2 ? 2u : 1u;
        st->n_need = s_need;
        st->cur_rank = cur_rank;
        st->npicks = s_npicks;
        st->iters = s_iters;
        st->n_wrows = s_nwrows; st->n_recount = s_nrecount; st->n_words = s_nwords;
    }
}

#include "setcover_batched.inc"
#include "setcover_flat.inc"

// ------------------------------------------------------------------------
// multi-launch solver (one gain launch + one apply launch per pick); used when
// the candidate sets are sharded over several GPUs
// ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
gain_kernel(const u64 *__restrict__ bm, const u32 *__restrict__ gs, const u32 *__restrict__ ge,
            const u32 *__restrict__ seg_row, const u32 *__restrict__ seg_univ,
            const u32 *__restrict__ set_seg_ptr, const u32 *__restrict__ left,
            const u32 *__restrict__ rank, const u32 *__restrict__ picked, u32 nsets, u32 nranks,
            u32 myrank, GreedyState *__restrict__ st) {
    __shared__ unsigned long long wkey[4];
    if (st->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 slot = blockIdx.x * 4 + wave;
    const u32 s = slot * nranks + myrank;
    unsigned long long key = 0;
    if (s < nsets && !picked[s] && rank[s] == st->cur_rank) {
        u32 g = 0;
        const u32 sb = set_seg_ptr[s], se = set_seg_ptr[s + 1];
        for (u32 q = sb + lane; q < se; q += WAVE) {
            u32 c = 0;
            for (u32 r = seg_row[q]; r < seg_row[q + 1]; ++r) c += range_popcount(bm, gs[r], ge[r]);
            u32 l = left[seg_univ[q]];
            g += c < l ? c : l;
        }
        unsigned long long g64 = g;
        for (int d = 32; d > 0; d >>= 1) g64 += __shfl_down(g64, d, WAVE);
        if (lane == 0 && g64 > 0) key = (g64 << ID_BITS) | (unsigned long long)(ID_MASK - s);
    }
    if (lane == 0) wkey[wave] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long k = wkey[0];
        k = wkey[1] > k ? wkey[1] : k;
        k = wkey[2] > k ? wkey[2] : k;
        k = wkey[3] > k ? wkey[3] : k;
        if (k) atomicMax(&st->best_key, k);
    }
}

__global__ void __launch_bounds__(256)
apply_kernel(unsigned long long *__restrict__ bm, const u32 *__restrict__ gs, const u32 *__restrict__ ge,
             const i32 *__restrict__ row_univ, const u32 *__restrict__ set_ptr,
             const u32 *__restrict__ seg_univ, const u32 *__restrict__ set_seg_ptr, u32 *__restrict__ usize,
             const u32 *__restrict__ can, u32 *__restrict__ left, u32 *__restrict__ picked,
             u32 *__restrict__ picks, GreedyState *__restrict__ st) {
    if (st->done) return;
    const unsigned long long key = st->best_key;
    __syncthreads();
    if ((key >> ID_BITS) == 0) {
        if (threadIdx.x == 0) {
            st->best_key = 0;
            st->iters++;
            st->cur_rank++;
            if (st->cur_rank >= st->nrank) st->done = 2;
        }
        return;
    }
    const u32 s = ID_MASK - (u32)(key & ID_MASK);
    for (u32 r = set_ptr[s] + threadIdx.x; r < set_ptr[s + 1]; r += blockDim.x) {
        u32 a = gs[r], e = ge[r];
        u32 w0 = a >> 6, w1 = (e - 1) >> 6;
        u32 cleared = 0;
        for (u32 w = w0; w <= w1; ++w) {
            u64 m = ~0ull;
            if (w == w0) m &= ~0ull << (a & 63);
            if (w == w1) m &= ~0ull >> (63 - ((e - 1) & 63));
            u64 old = atomicAnd(&bm[w], ~m);
            cleared += (u32)__popcll(old & m);
        }
        if (cleared) atomicSub(&usize[row_univ[r]], cleared);
    }
    __syncthreads();
    for (u32 q = set_seg_ptr[s] + threadIdx.x; q < set_seg_ptr[s + 1]; q += blockDim.x) {
        u32 u = seg_univ[q];
        u32 n = LD(&usize[u]);
        u32 c = can[u];
        u32 nl = n > c ? n - c : 0u;
        u32 ol = left[u];
        left[u] = nl;
        if (ol > 0 && nl == 0) atomicSub(&st->n_need, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        picked[s] = 1;
        picks[st->npicks] = s;
        st->npicks++;
        st->iters++;
        st->best_key = 0;
        if (LD(&st->n_need) == 0) st->done = 1;
    }
}

// ------------------------------------------------------------------------
// Which RCCL.  A process that imported torch first has torch's own librccl mapped
// under the same soname this library was linked against, and the loader would
// bind these calls to whichever copy came first.  The communicator therefore goes
// through ONE copy chosen here: the file named by CATCHHIP_RCCL_PATH, else
// /opt/rocm/lib/librccl.so, opened by path (a second, private copy if another
// file of that soname is already mapped -- a communicator never crosses copies);
// only if neither can be opened, whatever the loader bound.  catchhip_comm_info
// reports the version and the file actually in use.
#include <dlfcn.h>
namespace {
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string path, how;
    bool ready = false;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

const RcclApi &rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.ready) return g_rccl;
    void *h = nullptr;
    const char *want = getenv("CATCHHIP_RCCL_PATH");
    const char *tries[2] = {want && *want ? want : "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so"};
    for (int i = 0; i < 2 && !h; ++i) {
        h = dlopen(tries[i], RTLD_NOW | RTLD_LOCAL);
        if (h) g_rccl.how = std::string("dlopen ") + tries[i];
    }
    auto sym = [&](const char *name) -> void * { return h ? dlsym(h, name) : nullptr; };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl.GetVersion = (decltype(g_rccl.GetVersion))sym("ncclGetVersion");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.GetErrorString) {
        // the copy the loader bound this library's own references to
        g_rccl.GetUniqueId = &ncclGetUniqueId; g_rccl.CommInitRank = &ncclCommInitRank; g_rccl.CommDestroy = &ncclCommDestroy;
        g_rccl.AllReduce = &ncclAllReduce; g_rccl.GetErrorString = &ncclGetErrorString; g_rccl.GetVersion = &ncclGetVersion;
        g_rccl.how = "as linked (no RCCL file could be opened by path)";
    }
    Dl_info info;
    if (dladdr((void *)g_rccl.CommInitRank, &info) && info.dli_fname) g_rccl.path = info.dli_fname;
    g_rccl.ready = true;
    return g_rccl;
}
}  // namespace

extern "C" int catchhip_comm_info(char *buf, i64 len) {
    ARG_CHECK(buf && len > 0);
    const RcclApi &R = rccl();
    int v = 0;
    if (R.GetVersion) (void)R.GetVersion(&v);
    snprintf(buf, (size_t)len, "RCCL version code %d from %s (%s)", v, R.path.c_str(), R.how.c_str());
    return 0;
}

extern "C" int catchhip_comm_unique_id(u8 *id128) {
    ARG_CHECK(id128 != nullptr);
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) { chip_set_error("ncclGetUniqueId: %s", rccl().GetErrorString(r)); return CATCHHIP_ECOMM; }
    memcpy(id128, &id, 128);
    return 0;
}

extern "C" int catchhip_comm_init(catchhip_ctx *ctx, const u8 *id128, i32 nranks, i32 rank) {
    ARG_CHECK(ctx && id128 && nranks >= 1 && rank >= 0 && rank < nranks);
    HIP_TRY(hipSetDevice(ctx->device));
    (void)catchhip_comm_destroy(ctx);
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm;
    ncclResult_t r = rccl().CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) { chip_set_error("ncclCommInitRank: %s", rccl().GetErrorString(r)); return CATCHHIP_ECOMM; }
    ctx->comm = (void *)comm;
    ctx->nranks = nranks;
    ctx->rank = rank;
    return 0;
}

// fills a buffer with (rank + 1), SUM all-reduce, checks every element against nranks (nranks + 1) / 2
__global__ void __launch_bounds__(256)
comm_selftest_fill_kernel(u32 *buf, u32 n, u32 v) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = v;
}
__global__ void __launch_bounds__(256)
comm_selftest_check_kernel(const u32 *buf, u32 n, u32 want, u32 *bad) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && buf[i] != want) atomicAdd(bad, 1u);
}

extern "C" int catchhip_comm_selftest(catchhip_ctx *ctx, i64 nelem) {
    ARG_CHECK(ctx && nelem >= 1 && nelem < ((i64)1 << 30));
    if (!ctx->comm) { chip_set_error("comm_selftest: the context has no communicator"); return CATCHHIP_EINVAL; }
    PoolScope pool_scope(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DevBuf<u32> buf, bad;
    TRY(buf.alloc((size_t)nelem));
    TRY(bad.alloc(1));
    const u32 n = (u32)nelem;
    HIP_TRY(hipMemsetAsync(bad.p, 0, sizeof(u32), s));
    hipLaunchKernelGGL(comm_selftest_fill_kernel, dim3((unsigned)div_up((i64)n, 256)), dim3(256), 0, s, buf.p, n, (u32)ctx->rank + 1u);
    const ncclResult_t r = rccl().AllReduce(buf.p, buf.p, (size_t)n, ncclUint32, ncclSum, (ncclComm_t)ctx->comm, s);
    if (r != ncclSuccess) { chip_set_error("comm_selftest: ncclAllReduce: %s", rccl().GetErrorString(r)); return CATCHHIP_ECOMM; }
    const u32 want = (u32)((i64)ctx->nranks * (ctx->nranks + 1) / 2);
    hipLaunchKernelGGL(comm_selftest_check_kernel, dim3((unsigned)div_up((i64)n, 256)), dim3(256), 0, s, (const u32 *)buf.p, n, want, bad.p);
    u32 h_bad = 0;
    HIP_TRY(hipMemcpyAsync(&h_bad, bad.p, sizeof(u32), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (h_bad) { chip_set_error("comm_selftest: %u of %u elements differ from %u after the SUM all-reduce", h_bad, n, want); return CATCHHIP_ECOMM; }
    return 0;
}

extern "C" int catchhip_comm_destroy(catchhip_ctx *ctx) {
    if (ctx && ctx->comm) {
        (void)rccl().CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
        ctx->nranks = 1;
        ctx->rank = 0;
    }
    return 0;
}

// Host side of the frontier solver (setcover_batched.inc).  One zeroed arena,
// two set-up launches, rounds in batches, ONE synchronisation per batch that
// brings back the state, the picks and their keys through pinned memory.
static int greedy_frontier(catchhip_ctx *ctx, const catchhip_rows *R, u32 nsets, const u32 *h_rank, u32 nrank,
                           i64 *out_ids, i64 *n_out, int *retry) {
    hipStream_t s = ctx->stream;
    // deferred rows: R->n is the capacity of the table, its size is info[4]
    const u32 *d_info = R->deferred ? R->info.p : nullptr;
    const u32 nrows = (u32)R->n, nuniv = (u32)R->ngenomes;
    const size_t nwords = (size_t)(R->total / 64 + 2);
    // lanes per set (setcover_batched.inc): by the average rows per set, or --
    // when the row count is still on the device -- by the number of genomes
    const bool wide = d_info ? nuniv >= 256 : (i64)nrows >= 32 * (i64)nsets;
    // rows of more than 5 bitmap words take the lane-per-word kernels (a deferred
    // scan with such rows stops itself, info[5], and comes back through here)
    const bool long_rows = (!d_info && R->lmax > 257) || chip_test_env("CATCHHIP_GF_LONG") != nullptr;
    const u32 sets_per_wg = GF_THREADS / (wide ? 64 : 16);
    const unsigned gblocks = (unsigned)std::min<i64>(div_up(nsets, sets_per_wg), (i64)ctx->num_cus * 16);
    // arena: the zero-initialised part first
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_bm = take(8 * (nwords + 8)), o_ow0 = take(8 * (nwords + 8)), o_ow1 = take(8 * (nwords + 8)),
                 o_blk = take(16 * (size_t)gblocks), o_st = take(sizeof(GreedyState)),
                 o_picked = take(4 * (size_t)nsets), o_claimed = take(4 * (size_t)nsets),
                 o_lost = take(4 * (size_t)nsets), o_rank = take(4 * (size_t)nsets);
    const size_t zero_bytes = off;
    const size_t o_usize = take(4 * (size_t)nuniv), o_gain = take(4 * (size_t)nsets),
                 o_setptr = take(4 * ((size_t)nsets + 1)), o_frow = take(8 * (size_t)nrows),
                 o_flag = take(nrows), o_picks = take(4 * (size_t)nsets), o_keys = take(8 * (size_t)nsets);
    // live-set lists only where walking every set each round would dominate
    const bool use_list = nsets > 65536 && !chip_test_env("CATCHHIP_GF_NOLIST");
    const size_t o_live0 = use_list ? take(4 * (size_t)nsets) : 0, o_live1 = use_list ? take(4 * (size_t)nsets) : 0;
    DevBuf<u8> arena;
    TRY(arena.alloc(off));
    u8 *A = arena.p;
    // pinned staging: ranks up, {state, counters, picks, keys} down
    const size_t pin_bytes = sizeof(GreedyState) + 16 * (size_t)gblocks + 12 * (size_t)nsets + 64 + 64;
    TRY(chip_pinned_reserve(ctx, std::max(pin_bytes, 4 * (size_t)nsets)));
    HIP_TRY(hipMemsetAsync(A, 0, zero_bytes, s));
    if (h_rank) {
        memcpy(ctx->h_big, h_rank, 4 * (size_t)nsets);
        HIP_TRY(hipMemcpyAsync(A + o_rank, ctx->h_big, 4 * (size_t)nsets, hipMemcpyHostToDevice, s));
    }
    FrontArgs fa;
    fa.bm = (unsigned long long *)(A + o_bm);
    fa.owner[0] = (unsigned long long *)(A + o_ow0); fa.owner[1] = (unsigned long long *)(A + o_ow1);
    fa.frow = (const uint2 *)(A + o_frow); fa.row_univ = (const i32 *)R->univ.p; fa.set_ptr = (const u32 *)(A + o_setptr); fa.rank = (const u32 *)(A + o_rank);
    fa.usize = (u32 *)(A + o_usize); fa.gain = (u32 *)(A + o_gain); fa.claimed = (u32 *)(A + o_claimed);
    fa.lost = (u32 *)(A + o_lost);
    fa.picked = (u32 *)(A + o_picked); fa.picks = (u32 *)(A + o_picks);
    fa.pick_key = (unsigned long long *)(A + o_keys); fa.rowflag = A + o_flag;
    fa.blkcnt = (unsigned long long *)(A + o_blk); fa.st = (GreedyState *)(A + o_st);
    fa.nsets = nsets; fa.nwords = (u32)nwords;
    fa.live[0] = use_list ? (u32 *)(A + o_live0) : nullptr;
    fa.live[1] = use_list ? (u32 *)(A + o_live1) : nullptr;

    PhaseTimer tm(ctx, PHASE_GREEDY);
    hipLaunchKernelGGL(gf_build_kernel, dim3((unsigned)div_up(std::max(nrows, nsets + 1), 256)), dim3(256), 0, s,
                       (const i32 *)R->set_id.p, (const i32 *)R->univ.p, (const u32 *)R->gs.p, (const u32 *)R->ge.p,
                       nrows, d_info ? d_info + 4 : (const u32 *)nullptr, nsets, nrank, (uint2 *)(A + o_frow), fa.bm,
                       (u32 *)(A + o_setptr), fa.st);
    hipLaunchKernelGGL(gf_universe_kernel, dim3(nuniv), dim3(256), 0, s, (const unsigned long long *)fa.bm,
                       (const u32 *)R->genome_off.p, fa.usize, fa.st, d_info);
    tm.launch(2);
    // the host looks at the state after a batch of rounds (the kernels no-op
    // once everything is covered)
    const i64 max_rounds = (i64)nsets + nrank + 2;
    i64 rounds = 0;
    int per_sync = 10;
    u8 *H = (u8 *)ctx->h_big;
    GreedyState *h_st = (GreedyState *)H;
    unsigned long long *h_blk = (unsigned long long *)(H + sizeof(GreedyState));
    u32 *h_picks = (u32 *)(H + sizeof(GreedyState) + 16 * (size_t)gblocks);
    unsigned long long *h_keys = (unsigned long long *)(H + sizeof(GreedyState) + 16 * (size_t)gblocks +
                                                         ((4 * (size_t)nsets + 7) & ~(size_t)7));
    u32 *h_info = (u32 *)((u8 *)h_keys + 8 * (size_t)nsets);
    PhaseTimer tr(ctx, PHASE_GREEDY_ROUNDS);   // the round launches only
    for (;;) {
        for (int r = 0; r < per_sync; ++r, ++rounds) {
            if (long_rows) {
                if (wide) {
                    hipLaunchKernelGGL(gfl_count_claim_kernel<64>, dim3(gblocks), dim3(GF_THREADS), 0, s, fa, (u32)rounds);
                    hipLaunchKernelGGL(gfl_check_apply_kernel<64>, dim3(gblocks), dim3(GF_THREADS), 0, s, fa, (u32)rounds);
                } else {
                    hipLaunchKernelGGL(gfl_count_claim_kernel<16>, dim3(gblocks), dim3(GF_THREADS), 0, s, fa, (u32)rounds);
                    hipLaunchKernelGGL(gfl_check_apply_kernel<16>, dim3(gblocks), dim3(GF_THREADS), 0, s, fa, (u32)rounds);
                }
            } else if (wide) {
                hipLaunchKernelGGL(gf_count_claim_kernel<64>, dim3(gblocks), dim3(GF_THREADS), 0, s, fa, (u32)rounds);
                hipLaunchKernelGGL(gf_check_apply_kernel<64>, dim3(gblocks), dim3(GF_THREADS), 0, s, fa, (u32)rounds);
            } else {
                hipLaunchKernelGGL(gf_count_claim_kernel<16>, dim3(gblocks), dim3(GF_THREADS), 0, s, fa, (u32)rounds);
                hipLaunchKernelGGL(gf_check_apply_kernel<16>, dim3(gblocks), dim3(GF_THREADS), 0, s, fa, (u32)rounds);
            }
        }
        tm.launch(2 * per_sync);
        tr.stop();
        tm.stop();
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_st, fa.st, sizeof(GreedyState), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(h_blk, fa.blkcnt, 16 * (size_t)gblocks, hipMemcpyDeviceToHost, s));
        // small instances: picks and keys ride along (one synchronisation in all);
        // large ones: 12 B per candidate set per batch would be tens of MB, so
        // only the picks made are fetched, once the solver has finished
        const bool eager = nsets <= 65536;
        if (eager) {
            HIP_TRY(hipMemcpyAsync(h_picks, fa.picks, 4 * (size_t)nsets, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(h_keys, fa.pick_key, 8 * (size_t)nsets, hipMemcpyDeviceToHost, s));
        }
        if (d_info) HIP_TRY(hipMemcpyAsync(h_info, d_info, 16 * sizeof(u32), hipMemcpyDeviceToHost, s));
        const auto t_queued = std::chrono::steady_clock::now();
        HIP_TRY(hipStreamSynchronize(s));
        if (getenv("CATCHHIP_TIMING"))
            fprintf(stderr, "[catchhip]   solver batch: waited %.1f us for the device after queueing\n",
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_queued).count());
        if (h_st->done || h_st->n_need == 0) {
            if (!eager && h_st->npicks) {
                HIP_TRY(hipMemcpyAsync(h_picks, fa.picks, 4 * (size_t)h_st->npicks, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipMemcpyAsync(h_keys, fa.pick_key, 8 * (size_t)h_st->npicks, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
            }
            break;
        }
        if (rounds > max_rounds) { chip_set_error("setcover: round cap exceeded"); return CATCHHIP_EINVAL; }
        per_sync = 6;
        tr.stopped = false;   // further rounds extend both phases
        tm.stopped = false;
    }
    tr.launch(2 * rounds);
    tr.finish();
    tm.finish();
    if (d_info) {
        // the scan was never synchronised on its own: collect its timers and counters now
        chip_phase_collect(ctx, PHASE_SCAN);
        chip_phase_collect(ctx, PHASE_ROWS);
        ctx->counters[0] = h_info[2];
        ctx->counters[1] = h_info[9];
        ctx->seeds_dropped = (i64)h_info[9] - (i64)h_info[10];
        R->seed_ratio_seen = (double)h_info[9] / (double)std::max<i64>(R->total, 1);
        ctx->counters[7] = h_info[4];   // rows of the deferred table
        if (h_st->done == 3) { *retry = 1; return 0; }
    }
    i64 n_rec = 0, n_wrd = 0;
    for (unsigned b = 0; b < gblocks; ++b) { n_rec += (i64)h_blk[2 * b]; n_wrd += (i64)h_blk[2 * b + 1]; }
    ctx->phase_launches[PHASE_GREEDY] = h_st->iters;
    ctx->counters[2] = h_st->iters; ctx->counters[3] = h_st->npicks; ctx->counters[4] = 0;
    ctx->counters[5] = n_rec; ctx->counters[6] = n_wrd;
    if (h_st->done == 2) {
        chip_set_error("setcover: ranks exhausted while coverage is still required");
        return CATCHHIP_ERANK;
    }
    // sequential pick order = by rank, then by descending accept-time key
    // (see setcover_batched.inc)
    const u32 np = h_st->npicks;
    std::vector<u32> ord(np);
    for (u32 i = 0; i < np; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](u32 x, u32 y) {
        const u32 rx = h_rank ? h_rank[h_picks[x]] : 0u, ry = h_rank ? h_rank[h_picks[y]] : 0u;
        if (rx != ry) return rx < ry;
        return h_keys[x] > h_keys[y];
    });
    for (u32 i = 0; i < np; ++i) out_ids[i] = h_picks[ord[i]];
    *n_out = np;
    return 0;
}

// dense ranks: index into sorted(set(ranks.values())) (set_cover.py:353-354)
static u32 dense_ranks(const i64 *ranks, u32 nsets, std::vector<u32> &h_rank) {
    // (no ranks: h_rank stays empty -- a zeroed vector of 4 bytes per set was 16 MB of fresh pages per solve of S4's largest
    // group, 2.8 ms of host time between the row build and the solver's first launch in which the device sat idle)
    h_rank.clear();
    if (!ranks) return 1;
    h_rank.assign(nsets, 0);
    std::vector<i64> vals(ranks, ranks + nsets);
    std::sort(vals.begin(), vals.end());
    vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
    for (u32 i = 0; i < nsets; ++i)
        h_rank[i] = (u32)(std::lower_bound(vals.begin(), vals.end(), ranks[i]) - vals.begin());
    return (u32)vals.size();
}

int chip_greedy_deferred(catchhip_ctx *ctx, catchhip_rows *R, i64 num_sets, const i64 *ranks, i64 *out_ids,
                         i64 *n_out, int *retry) {
    *retry = 0;
    *n_out = 0;
    ctx->phase_ms[PHASE_CLAIM] = 0.0; ctx->phase_launches[PHASE_CLAIM] = 0;   // (only the row-parallel solver fills it)
    memset(ctx->solver_counters, 0, sizeof(ctx->solver_counters));
    if (num_sets <= 0 || num_sets >= (i64)ID_MASK || !R->deferred) { *retry = 1; return 0; }
    HIP_TRY(hipSetDevice(ctx->device));
    PoolScope pool_scope(ctx);
    std::vector<u32> h_rank;
    const u32 nrank = dense_ranks(ranks, (u32)num_sets, h_rank);
    int rc = greedy_frontier(ctx, R, (u32)num_sets, ranks ? h_rank.data() : nullptr, nrank, out_ids, n_out, retry);
    return rc;
}

extern "C" int catchhip_setcover_greedy(catchhip_ctx *ctx, const catchhip_rows *R, i64 num_sets, const i64 *ranks,
                                        const double *universe_p, i64 *out_ids, i64 *n_out) {
    ARG_CHECK(ctx && R && n_out && num_sets >= 0 && R->ctx == ctx);
    PoolScope pool_scope(ctx);
    *n_out = 0;
    ctx->phase_ms[PHASE_CLAIM] = 0.0; ctx->phase_launches[PHASE_CLAIM] = 0;   // (only the row-parallel solver fills it)
    memset(ctx->solver_counters, 0, sizeof(ctx->solver_counters));
    if (num_sets == 0 || R->n == 0) return 0;  // no universe has anything to cover
    ARG_CHECK(out_ids != nullptr);
    if (num_sets >= (i64)ID_MASK) { chip_set_error("setcover: more than 2^32-2 sets not supported"); return CATCHHIP_EINVAL; }
    if (R->n >= ((i64)1 << 31)) { chip_set_error("setcover: too many rows"); return CATCHHIP_EINVAL; }
    HIP_TRY(hipSetDevice(ctx->device));
    const u32 nrows = (u32)R->n, nsets = (u32)num_sets, nuniv = (u32)R->ngenomes;
    hipStream_t s = ctx->stream;
    // a communicator (even of one rank) selects the sharded multi-launch solver
    const bool distributed = ctx->comm != nullptr;

    std::vector<u32> h_rank;
    const u32 nrank = dense_ranks(ranks, nsets, h_rank);
    if (universe_p)
        for (u32 u = 0; u < nuniv; ++u)
            if (!(universe_p[u] >= 0.0 && universe_p[u] <= 1.0)) {
                chip_set_error("The coverage fraction (p) of each universe must be in [0,1]");
                return CATCHHIP_EINVAL;
            }

    // every universe fully covered: batched rounds (many independent picks per
    // round); otherwise one pick per iteration
    bool batched = !distributed && !chip_test_env("CATCHHIP_GREEDY_SEQUENTIAL");
    if (universe_p)
        for (u32 u = 0; u < nuniv && batched; ++u) batched = universe_p[u] == 1.0;

    int no_retry = 0;
    // large instances: the row-parallel, tile-ordered kernels (setcover_flat.inc).
    // Measured on S4 in round 2 (rows: rounds of the fused vs the flat kernels): 272 M: 65 vs
    // 30 ms; 44 M: 9.9 vs 6.7; 29 M: 6.6 vs 4.5; 19 M: 4.0 vs 3.4; 4 M: 1.5 vs 1.5 -- hence 2^22 rows until round 6.
    // Measured again on S3 x 0.25 / 0.5 / 1.0 with the kernels of round 6 (8-byte records, no count launch in round 0,
    // four rounds per read-back): 0.48 M rows: 1.56 vs 1.29 ms of solver kernels (step 6.9 vs 6.4 ms); 1.2 M: 2.98 vs
    // 1.68 (11.1 vs 8.0); 3.4 M (configs[2]): 5.96 vs 3.12 (20.1 vs 13.5 -- the fused family also reads back every
    // round).  So: 2^18 rows.  (Smaller instances never get here through the fused filter: below ~11 Mbases of
    // targets it queues scan and solve without a synchronisation, chip_greedy_deferred.)
    const i64 flat_min_rows = chip_test_env("CATCHHIP_FLAT_MIN_ROWS") ? atoll(chip_test_env("CATCHHIP_FLAT_MIN_ROWS")) : (i64)1 << 18;
    if (batched && R->lmax <= 257 && (i64)nrows >= flat_min_rows && nsets <= GR_MAX_SETS)
        return greedy_flat(ctx, R, nsets, ranks ? h_rank.data() : nullptr, nrank, out_ids, n_out);
    if (batched) return greedy_frontier(ctx, R, nsets, ranks ? h_rank.data() : nullptr, nrank, out_ids, n_out, &no_retry);
    // Partial coverage (some universe_p < 1) with rows of at most 257 bases: frontier rounds of the
    // row-parallel kernels with the universe test (setcover_flat.inc, "PARTIAL") whatever the size --
    // the one-workgroup solvers below take 3.9 ms per pick on S4's largest group (54.6 s for S4 under
    // -c 0.9 against 0.17 s under -c 1.0)
    if (universe_p && !distributed && R->lmax <= 257 && nsets <= GR_MAX_SETS && !chip_test_env("CATCHHIP_GREEDY_SEQUENTIAL") &&
        !chip_test_env("CATCHHIP_PARTIAL_SEQUENTIAL") && (i64)nrows >= (chip_test_env("CATCHHIP_PARTIAL_MIN_ROWS") ? atoll(chip_test_env("CATCHHIP_PARTIAL_MIN_ROWS")) : 0))
        return greedy_flat(ctx, R, nsets, ranks ? h_rank.data() : nullptr, nrank, out_ids, n_out, universe_p);

    DevBuf<u32> set_ptr, flag, idx, tmp, seg_row, seg_univ, seg_set, row_seg, set_seg_ptr, usize, can, left, rank,
        picked, picks;
    DevBuf<unsigned long long> bm;
    DevBuf<double> d_p;
    DevBuf<GreedyState> st;
    const size_t nwords = (size_t)(R->total / 64 + 2);
    TRY(set_ptr.alloc(nsets + 1));
    TRY(usize.alloc(nuniv));
    TRY(can.alloc(nuniv));
    TRY(left.alloc(nuniv));
    TRY(rank.alloc(nsets));
    TRY(picked.alloc(nsets));
    TRY(picks.alloc(nsets));
    TRY(bm.alloc(nwords));
    TRY(st.alloc(1));
    if (universe_p) {
        TRY(d_p.alloc(nuniv));
        HIP_TRY(hipMemcpyAsync(d_p.p, universe_p, sizeof(double) * nuniv, hipMemcpyHostToDevice, s));
    }
    if (ranks) HIP_TRY(hipMemcpyAsync(rank.p, h_rank.data(), sizeof(u32) * nsets, hipMemcpyHostToDevice, s));
    else HIP_TRY(hipMemsetAsync(rank.p, 0, sizeof(u32) * nsets, s));
    HIP_TRY(hipMemsetAsync(picked.p, 0, sizeof(u32) * nsets, s));
    HIP_TRY(hipMemsetAsync(bm.p, 0, sizeof(unsigned long long) * nwords, s));
    GreedyState h_st;
    memset(&h_st, 0, sizeof(h_st));
    h_st.nrank = nrank;
    HIP_TRY(hipMemcpyAsync(st.p, &h_st, sizeof(h_st), hipMemcpyHostToDevice, s));

    PhaseTimer tm(ctx, PHASE_GREEDY);
    const unsigned rb = (unsigned)div_up(nrows, 256), sb = (unsigned)div_up(nsets + 1, 256);
    hipLaunchKernelGGL(set_ptr_kernel, dim3(sb), dim3(256), 0, s, R->set_id.p, nrows, nsets, set_ptr.p);
    hipLaunchKernelGGL(bitmap_build_kernel, dim3(rb), dim3(256), 0, s, R->gs.p, R->ge.p, nrows, bm.p);
    hipLaunchKernelGGL(universe_size_kernel, dim3(nuniv), dim3(256), 0, s, (const u64 *)bm.p, R->genome_off.p,
                       nuniv, usize.p);
    hipLaunchKernelGGL(universe_need_kernel, dim3((unsigned)div_up(nuniv, 256)), dim3(256), 0, s, usize.p,
                       universe_p ? d_p.p : (const double *)nullptr, nuniv, can.p, left.p, st.p);
    hipLaunchKernelGGL(greedy_start_kernel, dim3(1), dim3(1), 0, s, st.p);
    tm.launch(5);
    HIP_TRY(hipGetLastError());

    u32 nseg = 0;
    if (!batched) {
        // (set, universe) segments of the row table
        TRY(flag.alloc(nrows));
        TRY(idx.alloc(nrows));
        TRY(row_seg.alloc(nrows));
        TRY(set_seg_ptr.alloc(nsets + 1));
        hipLaunchKernelGGL(seg_flag_kernel, dim3(rb), dim3(256), 0, s, R->set_id.p, R->univ.p, R->gs.p, R->ge.p, nrows,
                           flag.p, st.p);
        TRY(chip_exclusive_scan_u32(ctx, flag.p, idx.p, nrows, tmp));
        HIP_TRY(hipMemcpyAsync(ctx->h_pin, idx.p + (nrows - 1), sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync((u32 *)ctx->h_pin + 1, flag.p + (nrows - 1), sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        nseg = ((volatile u32 *)ctx->h_pin)[0] + ((volatile u32 *)ctx->h_pin)[1];
        TRY(seg_row.alloc(nseg + 1));
        TRY(seg_univ.alloc(nseg + 1));
        TRY(seg_set.alloc(nseg + 1));
        hipLaunchKernelGGL(seg_fill_kernel, dim3(rb), dim3(256), 0, s, flag.p, idx.p, R->set_id.p, R->univ.p, nrows, nseg,
                           seg_row.p, seg_univ.p, seg_set.p, row_seg.p);
        hipLaunchKernelGGL(set_seg_ptr_kernel, dim3(sb), dim3(256), 0, s, set_ptr.p, idx.p, nrows, nsets, nseg,
                           set_seg_ptr.p);
        tm.launch(6);
    }
    int rc = 0;
    if (!distributed) {
        // ---- persistent single-workgroup solver ---------------------------
        DevBuf<u32> prowcnt, segcnt, segcontrib, gain, dirty, pos_row, pos_row_alt, useg, useg_alt, useg_ptr,
            bucket;
        DevBuf<u64> pos_key, pos_key_alt, ukeys, ukeys_alt;
        DevBuf<uint4> pent, wrow;
        const u32 nbuckets = (u32)(R->total >> BUCKET_SHIFT) + 2;
        TRY(prowcnt.alloc(nrows));
        TRY(segcnt.alloc(nseg));
        TRY(segcontrib.alloc(nseg));
        TRY(gain.alloc(nsets));
        TRY(dirty.alloc(nrows));
        TRY(pos_key.alloc(nrows));
        TRY(pos_row.alloc(nrows));
        TRY(pent.alloc(nrows));
        TRY(wrow.alloc(nrows));
        TRY(bucket.alloc(nbuckets));
        TRY(ukeys.alloc(nseg));
        TRY(useg.alloc(nseg));
        TRY(useg_ptr.alloc(nuniv + 1));
        HIP_TRY(hipMemsetAsync(segcnt.p, 0, sizeof(u32) * nseg, s));
        hipLaunchKernelGGL(rowcnt_init_kernel, dim3(rb), dim3(256), 0, s, R->gs.p, R->ge.p, row_seg.p, nrows, segcnt.p);
        hipLaunchKernelGGL(seg_init_kernel, dim3((unsigned)div_up(nseg, 256)), dim3(256), 0, s, segcnt.p, seg_univ.p,
                           seg_set.p, left.p, nseg, segcontrib.p, st.p, ukeys.p, useg.p);
        hipLaunchKernelGGL(gain_init_kernel, dim3(sb), dim3(256), 0, s, segcontrib.p, set_seg_ptr.p, nsets, gain.p);
        hipLaunchKernelGGL(wrow_fill_kernel, dim3(rb), dim3(256), 0, s, R->gs.p, R->ge.p, R->set_id.p, R->univ.p,
                           row_seg.p, can.p, nrows, wrow.p);
        hipLaunchKernelGGL(pos_key_kernel, dim3(rb), dim3(256), 0, s, R->gs.p, nrows, pos_key.p, pos_row.p);
        TRY(chip_radix_sort_pairs(ctx, pos_key, pos_key_alt, pos_row, pos_row_alt, nrows,
                                  std::max(1, ceil_log2_u64((u64)R->total + 1))));
        hipLaunchKernelGGL(pent_fill_kernel, dim3(rb), dim3(256), 0, s, pos_row.p, R->gs.p, R->ge.p, R->set_id.p,
                           row_seg.p, nrows, pent.p, prowcnt.p);
        hipLaunchKernelGGL(bucket_kernel, dim3((unsigned)div_up(nbuckets, 256)), dim3(256), 0, s, pos_key.p, nrows,
                           nbuckets, BUCKET_SHIFT, bucket.p);
        TRY(chip_radix_sort_pairs(ctx, ukeys, ukeys_alt, useg, useg_alt, nseg,
                                  std::max(1, ceil_log2_u64((u64)nuniv + 1))));
        hipLaunchKernelGGL(useg_ptr_kernel, dim3((unsigned)div_up(nuniv + 1, 256)), dim3(256), 0, s, ukeys.p, nseg,
                           nuniv, useg_ptr.p);
        GreedyArgs a;
        a.bm = bm.p; a.wrow = wrow.p; a.set_ptr = set_ptr.p; a.set_seg_ptr = set_seg_ptr.p;
        a.seg_univ = seg_univ.p; a.seg_set = seg_set.p; a.pent = pent.p; a.prowcnt = prowcnt.p;
        a.bucket = bucket.p; a.useg_ptr = useg_ptr.p; a.useg = useg.p; a.can = can.p;
        a.rank = rank.p; a.usize = usize.p; a.left = left.p; a.segcnt = segcnt.p; a.segcontrib = segcontrib.p;
        a.gain = gain.p; a.picked = picked.p; a.picks = picks.p; a.dirty = dirty.p; a.st = st.p;
        a.nrows = nrows; a.nsets = nsets; a.nuniv = nuniv; a.chunk = (u32)div_up(nsets, GW_THREADS);
        hipLaunchKernelGGL(greedy_wg_kernel, dim3(1), dim3(GW_THREADS), 0, s, a);
        tm.launch(6);
        tm.stop();
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&h_st, st.p, sizeof(h_st), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        tm.finish();
        ctx->phase_launches[PHASE_GREEDY] = h_st.iters;  // greedy iterations inside the persistent launch
        ctx->counters[2] = h_st.iters; ctx->counters[3] = h_st.npicks; ctx->counters[4] = (i64)h_st.n_wrows;
        ctx->counters[5] = (i64)h_st.n_recount; ctx->counters[6] = (i64)h_st.n_words;
#ifdef CATCHHIP_PROFILE
        if (chip_test_env("CATCHHIP_PROF")) {
            fprintf(stderr, "[catchhip] greedy wg: iters=%u picks=%u ms=%.3f ticks/iter:", h_st.iters, h_st.npicks,
                    ctx->phase_ms[PHASE_GREEDY]);
            for (int i = 0; i < 4; ++i) fprintf(stderr, " p%d=%.0f", i, (double)h_st.prof[i] / (h_st.iters ? h_st.iters : 1));
            fprintf(stderr, "\n");
        }
#endif
    } else {
        const u32 nranks = (u32)ctx->nranks, myrank = (u32)ctx->rank;
        const u32 my_sets = (u32)div_up((i64)nsets, nranks);
        const unsigned gb = (unsigned)div_up(my_sets, 4);
        const int BATCH = 64;
        const i64 max_iters = (i64)nsets + nrank + 2;
        i64 issued = 0;
        for (;;) {
            for (int b = 0; b < BATCH; ++b) {
                hipLaunchKernelGGL(gain_kernel, dim3(gb), dim3(256), 0, s, (const u64 *)bm.p, R->gs.p, R->ge.p,
                                   seg_row.p, seg_univ.p, set_seg_ptr.p, left.p, rank.p, picked.p, nsets, nranks,
                                   myrank, st.p);
                ncclResult_t r = rccl().AllReduce(&st.p->best_key, &st.p->best_key, 1, ncclUint64, ncclMax,
                                                  (ncclComm_t)ctx->comm, s);
                if (r != ncclSuccess) { chip_set_error("ncclAllReduce: %s", rccl().GetErrorString(r)); return CATCHHIP_ECOMM; }
                hipLaunchKernelGGL(apply_kernel, dim3(1), dim3(256), 0, s, bm.p, R->gs.p, R->ge.p, R->univ.p,
                                   set_ptr.p, seg_univ.p, set_seg_ptr.p, usize.p, can.p, left.p, picked.p, picks.p,
                                   st.p);
            }
            tm.launch(2 * BATCH);
            issued += BATCH;
            HIP_TRY(hipMemcpyAsync(&h_st, st.p, sizeof(h_st), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipGetLastError());
            if (h_st.done) break;
            if (issued > max_iters) { chip_set_error("setcover: iteration cap exceeded"); rc = CATCHHIP_EINVAL; break; }
        }
        tm.stop();
        tm.finish();
    }
    if (rc) return rc;
    if (h_st.done == 2) {
        chip_set_error("setcover: ranks exhausted while coverage is still required");
        return CATCHHIP_ERANK;
    }
    if (h_st.done != 1) { chip_set_error("setcover: solver did not finish"); return CATCHHIP_EHIP; }
    std::vector<u32> h_picks(h_st.npicks);
    if (h_st.npicks) {
        HIP_TRY(hipMemcpyAsync(h_picks.data(), picks.p, sizeof(u32) * h_st.npicks, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    for (u32 i = 0; i < h_st.npicks; ++i) out_ids[i] = h_picks[i];
    *n_out = h_st.npicks;
    return 0;
}

#include "setcover_sharded.inc"
