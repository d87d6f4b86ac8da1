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
// Single-GPU solver: after grid-wide set-up kernels, ONE persistent workgroup
// runs the whole greedy loop (the picks are inherently sequential; a
// workgroup barrier costs a fraction of a microsecond where a kernel boundary
// costs several).  Per pick it (1) takes the arg-max of a two-level max
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
    u32 pad;
    unsigned long long prof[8];  // shader-clock ticks per phase (thread 0)
};

#define ID_BITS 24
#define ID_MASK 0xFFFFFFu
#define GAIN_BLOCK 256  // sets per block of the two-level max structure

#define LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define PROF(i) do { if (tid == 0) { unsigned long long t_ = __builtin_readcyclecounter(); st->prof[i] += t_ - t_prev; t_prev = t_; } } while (0)

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
__device__ __forceinline__ u32 range_popcount_l2(const unsigned long long *bm, u32 s, u32 e) {
    u32 w0 = s >> 6, w1 = (e - 1) >> 6;
    u64 m0 = ~0ull << (s & 63);
    u64 m1 = ~0ull >> (63 - ((e - 1) & 63));
    if (w0 == w1) return (u32)__popcll(LD(&bm[w0]) & m0 & m1);
    u32 c = (u32)__popcll(LD(&bm[w0]) & m0);
    for (u32 w = w0 + 1; w < w1; ++w) c += (u32)__popcll(LD(&bm[w]));
    return c + (u32)__popcll(LD(&bm[w1]) & m1);
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
                   u32 nrows, u32 *__restrict__ rowcnt, u32 *__restrict__ segcnt) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    u32 c = ge[r] - gs[r];
    rowcnt[r] = c;
    atomicAdd(&segcnt[row_seg[r]], c);
}

__global__ void __launch_bounds__(256)
seg_init_kernel(const u32 *__restrict__ segcnt, const u32 *__restrict__ seg_univ,
                const u32 *__restrict__ seg_set, const u32 *__restrict__ left, u32 nseg,
                u32 *__restrict__ segcontrib, u32 *__restrict__ gain, u32 *__restrict__ segmax,
                u64 *__restrict__ ukeys, u32 *__restrict__ uvals) {
    u32 q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nseg) return;
    u32 u = seg_univ[q], c = segcnt[q], l = left[u];
    u32 contrib = c < l ? c : l;
    segcontrib[q] = contrib;
    atomicAdd(&gain[seg_set[q]], contrib);
    atomicMax(&segmax[u], c);
    ukeys[q] = u;   // for the per-universe segment index
    uvals[q] = q;
}

__global__ void __launch_bounds__(256)
pos_key_kernel(const u32 *__restrict__ gs, u32 nrows, u64 *__restrict__ keys, u32 *__restrict__ vals) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    keys[r] = gs[r];
    vals[r] = r;
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
    const u32 *gs, *ge;
    const i32 *row_set, *row_univ;
    const u32 *row_seg;
    const u32 *set_ptr, *set_seg_ptr;
    const u32 *seg_univ, *seg_set;
    const u64 *pos_key;   // sorted row starts
    const u32 *pos_row;   // row index per sorted slot
    const u32 *useg_ptr, *useg;
    const u32 *can, *segmax, *rank;
    u32 *usize, *left, *rowcnt, *segcnt, *segcontrib, *gain;
    u32 *picked, *picks, *ubind, *blockdirty, *dirty;
    unsigned long long *blockmax;
    GreedyState *st;
    u32 nrows, nsets, nuniv, nblocks;
};

#define GW_THREADS 1024
#define GW_WAVES (GW_THREADS / WAVE)
#define GW_MAXBIND 2048

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        unsigned long long o = __shfl_down(v, d, WAVE);
        v = o > v ? o : v;
    }
    return v;
}

// key of block b over the sets of the current rank that are not picked
__device__ __forceinline__ void refresh_block(const GreedyArgs &a, u32 b, u32 cur_rank, int lane) {
    unsigned long long k = 0;
    const u32 base = b * GAIN_BLOCK;
#pragma unroll
    for (int j = 0; j < GAIN_BLOCK / WAVE; ++j) {
        u32 s = base + j * WAVE + lane;
        if (s < a.nsets && a.rank[s] == cur_rank && !LD(&a.picked[s])) {
            unsigned long long g = LD(&a.gain[s]);
            if (g) {
                unsigned long long kk = (g << ID_BITS) | (unsigned long long)(ID_MASK - s);
                k = kk > k ? kk : k;
            }
        }
    }
    k = wave_max_u64(k);
    if (lane == 0) { ST(&a.blockmax[b], k); ST(&a.blockdirty[b], 0u); }
}

__global__ void __launch_bounds__(GW_THREADS)
greedy_wg_kernel(GreedyArgs a) {
    __shared__ unsigned long long s_red[GW_WAVES];
    __shared__ unsigned long long s_key;
    __shared__ u32 s_bind[GW_MAXBIND];
    __shared__ u32 s_nbind, s_ndirty, s_need, s_rank, s_stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    GreedyState *st = a.st;

    if (tid == 0) { s_need = st->n_need; s_rank = st->cur_rank; s_stop = st->done; s_nbind = 0; }
    __syncthreads();
    if (s_stop) return;
    for (u32 b = wave; b < a.nblocks; b += GW_WAVES) refresh_block(a, b, s_rank, lane);
    DRAIN();
    __syncthreads();
    unsigned long long t_prev = __builtin_readcyclecounter();

    for (;;) {
        // ---- arg-max over the block maxima --------------------------------
        unsigned long long k = 0;
        for (u32 b = tid; b < a.nblocks; b += GW_THREADS) {
            unsigned long long v = LD(&a.blockmax[b]);
            k = v > k ? v : k;
        }
        k = wave_max_u64(k);
        if (lane == 0) s_red[wave] = k;
        __syncthreads();
        if (tid == 0) {
            unsigned long long m = 0;
            for (int w = 0; w < GW_WAVES; ++w) m = s_red[w] > m ? s_red[w] : m;
            s_key = m;
            st->iters++;
            if ((m >> ID_BITS) == 0) {
                // no set of this rank covers anything still needed: next rank
                // (set_cover.py:522-526)
                s_rank++;
                st->cur_rank = s_rank;
                if (s_rank >= st->nrank) { st->done = 2; s_stop = 1; }
            } else {
                u32 s = ID_MASK - (u32)(m & ID_MASK);
                ST(&a.picked[s], 1u);
                a.picks[st->npicks] = s;
                st->npicks++;
            }
            s_nbind = 0;
            s_ndirty = 0;
        }
        __syncthreads(); PROF(0);
        if (s_stop) return;
        const unsigned long long key = s_key;
        if ((key >> ID_BITS) == 0) {
            for (u32 b = wave; b < a.nblocks; b += GW_WAVES) refresh_block(a, b, s_rank, lane);
            DRAIN();
            __syncthreads();
            continue;
        }
        const u32 s = ID_MASK - (u32)(key & ID_MASK);
        const u32 r0 = a.set_ptr[s], r1 = a.set_ptr[s + 1];

        // ---- apply: remove the winner's elements (set_cover.py:531-550) ----
        for (u32 r = r0 + tid; r < r1; r += GW_THREADS) {
            u32 x = a.gs[r], e = a.ge[r];
            u32 w0 = x >> 6, w1 = (e - 1) >> 6;
            u32 cleared = 0;
            for (u32 w = w0; w <= w1; ++w) {
                u64 m = ~0ull;
                if (w == w0) m &= ~0ull << (x & 63);
                if (w == w1) m &= ~0ull >> (63 - ((e - 1) & 63));
                u64 old = atomicAnd(&a.bm[w], ~m);
                cleared += (u32)__popcll(old & m);
            }
            if (cleared) atomicSub(&a.usize[a.row_univ[r]], cleared);
        }
        DRAIN();
        __syncthreads(); PROF(1);
        for (u32 q = a.set_seg_ptr[s] + tid; q < a.set_seg_ptr[s + 1]; q += GW_THREADS) {
            u32 u = a.seg_univ[q];
            u32 n = LD(&a.usize[u]);
            u32 c = a.can[u];
            u32 nl = n > c ? n - c : 0u;
            u32 ol = LD(&a.left[u]);
            if (nl != ol) {
                ST(&a.left[u], nl);
                if (ol > 0 && nl == 0) atomicSub(&s_need, 1u);
                // min(left, count) can only bind when part of the universe may
                // stay uncovered and the need dropped below the largest count
                if (c > 0 && nl < a.segmax[u]) {
                    u32 slot = atomicAdd(&s_nbind, 1u);
                    if (slot < GW_MAXBIND) s_bind[slot] = u;
                    ST(&a.ubind[u], 1u);
                }
            }
        }
        DRAIN();
        __syncthreads(); PROF(2);

        // ---- re-count the rows that overlap a cleared range ----------------
        const u32 lmax = st->lmax;
        for (u32 wr = r0 + wave; wr < r1; wr += GW_WAVES) {
            const u32 x = a.gs[wr], e = a.ge[wr];
            const u32 lo_start = x > lmax ? x - lmax : 0u;  // rows starting before this cannot reach x
            u32 lo = 0, hi = a.nrows;
            while (lo < hi) {
                u32 mid = (lo + hi) >> 1;
                if ((u32)a.pos_key[mid] < lo_start) lo = mid + 1; else hi = mid;
            }
            for (u32 y = lo + lane; y < a.nrows; y += WAVE) {
                if ((u32)a.pos_key[y] >= e) break;
                const u32 r = a.pos_row[y];
                if (a.ge[r] <= x) continue;
                const u32 sr = (u32)a.row_set[r];
                if (LD(&a.picked[sr])) continue;
                const u32 nc = range_popcount_l2(a.bm, a.gs[r], a.ge[r]);
                const u32 oc = atomicExch(&a.rowcnt[r], nc);  // a row reached from two ranges is patched once
                if (oc != nc) {
                    const u32 q = a.row_seg[r];
                    atomicSub(&a.segcnt[q], oc - nc);
                    a.dirty[atomicAdd(&s_ndirty, 1u)] = q;   // <= one entry per row per pick
                }
            }
        }
        DRAIN();
        __syncthreads(); PROF(3);
        // contribution min(left, count) of every segment whose count changed
        {
            const u32 nd = s_ndirty;
            for (u32 i = tid; i < nd; i += GW_THREADS) {
                const u32 q = a.dirty[i];
                const u32 ss = a.seg_set[q];
                const u32 l = LD(&a.left[a.seg_univ[q]]), c = LD(&a.segcnt[q]);
                const u32 nc = c < l ? c : l;
                const u32 oc = atomicExch(&a.segcontrib[q], nc);
                if (oc != nc) { atomicAdd(&a.gain[ss], nc - oc); ST(&a.blockdirty[ss / GAIN_BLOCK], 1u); }
            }
        }
        DRAIN();
        __syncthreads(); PROF(4);

        // ---- universes whose need became binding: re-evaluate min(left, count)
        {
            const u32 nb = s_nbind;
            if (nb > GW_MAXBIND) {
                // overflow of the LDS list: walk every universe's flag instead
                for (u32 u = wave; u < a.nuniv; u += GW_WAVES) {
                    if (!LD(&a.ubind[u])) continue;
                    const u32 l = LD(&a.left[u]);
                    for (u32 z = a.useg_ptr[u] + lane; z < a.useg_ptr[u + 1]; z += WAVE) {
                        u32 q = a.useg[z], ss = a.seg_set[q];
                        if (LD(&a.picked[ss])) continue;
                        u32 c = LD(&a.segcnt[q]);
                        u32 nc = c < l ? c : l;
                        u32 oc = atomicExch(&a.segcontrib[q], nc);
                        if (oc != nc) { atomicAdd(&a.gain[ss], nc - oc); ST(&a.blockdirty[ss / GAIN_BLOCK], 1u); }
                    }
                    if (lane == 0) ST(&a.ubind[u], 0u);
                }
            } else {
                for (u32 i = wave; i < nb; i += GW_WAVES) {
                    const u32 u = s_bind[i];
                    const u32 l = LD(&a.left[u]);
                    for (u32 z = a.useg_ptr[u] + lane; z < a.useg_ptr[u + 1]; z += WAVE) {
                        u32 q = a.useg[z], ss = a.seg_set[q];
                        if (LD(&a.picked[ss])) continue;
                        u32 c = LD(&a.segcnt[q]);
                        u32 nc = c < l ? c : l;
                        u32 oc = atomicExch(&a.segcontrib[q], nc);
                        if (oc != nc) { atomicAdd(&a.gain[ss], nc - oc); ST(&a.blockdirty[ss / GAIN_BLOCK], 1u); }
                    }
                    if (lane == 0) ST(&a.ubind[u], 0u);
                }
            }
        }
        if (tid == 0) ST(&a.blockdirty[s / GAIN_BLOCK], 1u);
        DRAIN();
        __syncthreads(); PROF(5);

        // ---- refresh dirty blocks of the max structure ---------------------
        for (u32 base = wave * WAVE; base < a.nblocks; base += GW_WAVES * WAVE) {
            u32 b = base + lane;
            bool dirty = (b < a.nblocks) && LD(&a.blockdirty[b]);
            u64 mask = __ballot(dirty);
            while (mask) {
                int j = __ffsll((unsigned long long)mask) - 1;
                mask &= mask - 1;
                refresh_block(a, base + j, s_rank, lane);
            }
        }
        DRAIN();
        __syncthreads(); PROF(6);
        if (tid == 0 && s_need == 0) { st->done = 1; st->n_need = 0; s_stop = 1; }
        __syncthreads();
        if (s_stop) return;
    }
}

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
extern "C" int catchhip_comm_unique_id(u8 *id128) {
    ARG_CHECK(id128 != nullptr);
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { chip_set_error("ncclGetUniqueId: %s", ncclGetErrorString(r)); return CATCHHIP_ECOMM; }
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
    ncclResult_t r = ncclCommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) { chip_set_error("ncclCommInitRank: %s", ncclGetErrorString(r)); return CATCHHIP_ECOMM; }
    ctx->comm = (void *)comm;
    ctx->nranks = nranks;
    ctx->rank = rank;
    return 0;
}

extern "C" int catchhip_comm_destroy(catchhip_ctx *ctx) {
    if (ctx && ctx->comm) {
        (void)ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
        ctx->nranks = 1;
        ctx->rank = 0;
    }
    return 0;
}

extern "C" int catchhip_setcover_greedy(catchhip_ctx *ctx, const catchhip_rows *R, i64 num_sets, const i64 *ranks,
                                        const double *universe_p, i64 *out_ids, i64 *n_out) {
    ARG_CHECK(ctx && R && n_out && num_sets >= 0 && R->ctx == ctx);
    *n_out = 0;
    if (num_sets == 0 || R->n == 0) return 0;  // no universe has anything to cover
    ARG_CHECK(out_ids != nullptr);
    if (num_sets >= (i64)ID_MASK) { chip_set_error("setcover: more than 2^24-1 sets not supported"); return CATCHHIP_EINVAL; }
    if (R->n >= ((i64)1 << 31)) { chip_set_error("setcover: too many rows"); return CATCHHIP_EINVAL; }
    HIP_TRY(hipSetDevice(ctx->device));
    const u32 nrows = (u32)R->n, nsets = (u32)num_sets, nuniv = (u32)R->ngenomes;
    hipStream_t s = ctx->stream;
    const bool distributed = ctx->comm != nullptr && ctx->nranks > 1;

    // dense ranks: index into sorted(set(ranks.values())) (set_cover.py:353-354)
    std::vector<u32> h_rank(nsets, 0);
    u32 nrank = 1;
    if (ranks) {
        std::vector<i64> vals(ranks, ranks + nsets);
        std::sort(vals.begin(), vals.end());
        vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
        nrank = (u32)vals.size();
        for (u32 i = 0; i < nsets; ++i)
            h_rank[i] = (u32)(std::lower_bound(vals.begin(), vals.end(), ranks[i]) - vals.begin());
    }
    if (universe_p)
        for (u32 u = 0; u < nuniv; ++u)
            if (!(universe_p[u] >= 0.0 && universe_p[u] <= 1.0)) {
                chip_set_error("The coverage fraction (p) of each universe must be in [0,1]");
                return CATCHHIP_EINVAL;
            }

    DevBuf<u32> set_ptr, flag, idx, tmp, seg_row, seg_univ, seg_set, row_seg, set_seg_ptr, usize, can, left, rank,
        picked, picks;
    DevBuf<unsigned long long> bm;
    DevBuf<double> d_p;
    DevBuf<GreedyState> st;
    const size_t nwords = (size_t)(R->total / 64 + 2);
    TRY(set_ptr.alloc(nsets + 1));
    TRY(flag.alloc(nrows));
    TRY(idx.alloc(nrows));
    TRY(row_seg.alloc(nrows));
    TRY(set_seg_ptr.alloc(nsets + 1));
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
    HIP_TRY(hipMemcpyAsync(rank.p, h_rank.data(), sizeof(u32) * nsets, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(picked.p, 0, sizeof(u32) * nsets, s));
    HIP_TRY(hipMemsetAsync(bm.p, 0, sizeof(unsigned long long) * nwords, s));
    GreedyState h_st;
    memset(&h_st, 0, sizeof(h_st));
    h_st.nrank = nrank;
    HIP_TRY(hipMemcpyAsync(st.p, &h_st, sizeof(h_st), hipMemcpyHostToDevice, s));

    PhaseTimer tm(ctx, PHASE_GREEDY);
    const unsigned rb = (unsigned)div_up(nrows, 256), sb = (unsigned)div_up(nsets + 1, 256);
    hipLaunchKernelGGL(set_ptr_kernel, dim3(sb), dim3(256), 0, s, R->set_id.p, nrows, nsets, set_ptr.p);
    hipLaunchKernelGGL(seg_flag_kernel, dim3(rb), dim3(256), 0, s, R->set_id.p, R->univ.p, R->gs.p, R->ge.p, nrows,
                       flag.p, st.p);
    TRY(chip_exclusive_scan_u32(ctx, flag.p, idx.p, nrows, tmp));
    HIP_TRY(hipMemcpyAsync(ctx->h_pin, idx.p + (nrows - 1), sizeof(u32), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync((u32 *)ctx->h_pin + 1, flag.p + (nrows - 1), sizeof(u32), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const u32 nseg = ((volatile u32 *)ctx->h_pin)[0] + ((volatile u32 *)ctx->h_pin)[1];
    TRY(seg_row.alloc(nseg + 1));
    TRY(seg_univ.alloc(nseg + 1));
    TRY(seg_set.alloc(nseg + 1));
    hipLaunchKernelGGL(seg_fill_kernel, dim3(rb), dim3(256), 0, s, flag.p, idx.p, R->set_id.p, R->univ.p, nrows, nseg,
                       seg_row.p, seg_univ.p, seg_set.p, row_seg.p);
    hipLaunchKernelGGL(set_seg_ptr_kernel, dim3(sb), dim3(256), 0, s, set_ptr.p, idx.p, nrows, nsets, nseg,
                       set_seg_ptr.p);
    hipLaunchKernelGGL(bitmap_build_kernel, dim3(rb), dim3(256), 0, s, R->gs.p, R->ge.p, nrows, bm.p);
    hipLaunchKernelGGL(universe_size_kernel, dim3(nuniv), dim3(256), 0, s, (const u64 *)bm.p, R->genome_off.p,
                       nuniv, usize.p);
    hipLaunchKernelGGL(universe_need_kernel, dim3((unsigned)div_up(nuniv, 256)), dim3(256), 0, s, usize.p,
                       universe_p ? d_p.p : (const double *)nullptr, nuniv, can.p, left.p, st.p);
    hipLaunchKernelGGL(greedy_start_kernel, dim3(1), dim3(1), 0, s, st.p);
    tm.launch(8);
    HIP_TRY(hipGetLastError());

    int rc = 0;
    if (!distributed) {
        // ---- persistent single-workgroup solver ---------------------------
        DevBuf<u32> rowcnt, segcnt, segcontrib, gain, segmax, ubind, blockdirty, dirty, pos_row, pos_row_alt, useg,
            useg_alt, useg_ptr;
        DevBuf<u64> pos_key, pos_key_alt, ukeys, ukeys_alt;
        DevBuf<unsigned long long> blockmax;
        const u32 nblocks = (u32)div_up(nsets, GAIN_BLOCK);
        TRY(rowcnt.alloc(nrows));
        TRY(segcnt.alloc(nseg));
        TRY(segcontrib.alloc(nseg));
        TRY(gain.alloc(nsets));
        TRY(segmax.alloc(nuniv));
        TRY(ubind.alloc(nuniv));
        TRY(blockdirty.alloc(nblocks));
        TRY(dirty.alloc(nrows));
        TRY(blockmax.alloc(nblocks));
        TRY(pos_key.alloc(nrows));
        TRY(pos_row.alloc(nrows));
        TRY(ukeys.alloc(nseg));
        TRY(useg.alloc(nseg));
        TRY(useg_ptr.alloc(nuniv + 1));
        HIP_TRY(hipMemsetAsync(segcnt.p, 0, sizeof(u32) * nseg, s));
        HIP_TRY(hipMemsetAsync(gain.p, 0, sizeof(u32) * nsets, s));
        HIP_TRY(hipMemsetAsync(segmax.p, 0, sizeof(u32) * nuniv, s));
        HIP_TRY(hipMemsetAsync(ubind.p, 0, sizeof(u32) * nuniv, s));
        HIP_TRY(hipMemsetAsync(blockdirty.p, 0, sizeof(u32) * nblocks, s));
        hipLaunchKernelGGL(rowcnt_init_kernel, dim3(rb), dim3(256), 0, s, R->gs.p, R->ge.p, row_seg.p, nrows,
                           rowcnt.p, segcnt.p);
        hipLaunchKernelGGL(seg_init_kernel, dim3((unsigned)div_up(nseg, 256)), dim3(256), 0, s, segcnt.p, seg_univ.p,
                           seg_set.p, left.p, nseg, segcontrib.p, gain.p, segmax.p, ukeys.p, useg.p);
        hipLaunchKernelGGL(pos_key_kernel, dim3(rb), dim3(256), 0, s, R->gs.p, nrows, pos_key.p, pos_row.p);
        TRY(chip_radix_sort_pairs(ctx, pos_key, pos_key_alt, pos_row, pos_row_alt, nrows,
                                  std::max(1, ceil_log2_u64((u64)R->total + 1))));
        TRY(chip_radix_sort_pairs(ctx, ukeys, ukeys_alt, useg, useg_alt, nseg,
                                  std::max(1, ceil_log2_u64((u64)nuniv + 1))));
        hipLaunchKernelGGL(useg_ptr_kernel, dim3((unsigned)div_up(nuniv + 1, 256)), dim3(256), 0, s, ukeys.p, nseg,
                           nuniv, useg_ptr.p);
        GreedyArgs a;
        a.bm = bm.p; a.gs = R->gs.p; a.ge = R->ge.p; a.row_set = R->set_id.p; a.row_univ = R->univ.p;
        a.row_seg = row_seg.p; a.set_ptr = set_ptr.p; a.set_seg_ptr = set_seg_ptr.p; a.seg_univ = seg_univ.p;
        a.seg_set = seg_set.p; a.pos_key = pos_key.p; a.pos_row = pos_row.p; a.useg_ptr = useg_ptr.p;
        a.useg = useg.p; a.can = can.p; a.segmax = segmax.p; a.rank = rank.p; a.usize = usize.p; a.left = left.p;
        a.rowcnt = rowcnt.p; a.segcnt = segcnt.p; a.segcontrib = segcontrib.p; a.gain = gain.p; a.picked = picked.p;
        a.picks = picks.p; a.ubind = ubind.p; a.blockdirty = blockdirty.p; a.dirty = dirty.p; a.blockmax = blockmax.p; a.st = st.p;
        a.nrows = nrows; a.nsets = nsets; a.nuniv = nuniv; a.nblocks = nblocks;
        hipLaunchKernelGGL(greedy_wg_kernel, dim3(1), dim3(GW_THREADS), 0, s, a);
        tm.launch(6);
        tm.stop();
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&h_st, st.p, sizeof(h_st), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        tm.finish();
        ctx->phase_launches[PHASE_GREEDY] = h_st.iters;  // greedy iterations inside the persistent launch
        if (getenv("CATCHHIP_PROF")) {
            fprintf(stderr, "[catchhip] greedy wg: iters=%u picks=%u ms=%.3f ticks/iter:", h_st.iters, h_st.npicks,
                    ctx->phase_ms[PHASE_GREEDY]);
            for (int i = 0; i < 7; ++i) fprintf(stderr, " p%d=%.0f", i, (double)h_st.prof[i] / (h_st.iters ? h_st.iters : 1));
            fprintf(stderr, "\n");
        }
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
                ncclResult_t r = ncclAllReduce(&st.p->best_key, &st.p->best_key, 1, ncclUint64, ncclMax,
                                               (ncclComm_t)ctx->comm, s);
                if (r != ncclSuccess) { chip_set_error("ncclAllReduce: %s", ncclGetErrorString(r)); return CATCHHIP_ECOMM; }
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
