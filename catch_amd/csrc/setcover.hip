// K2: greedy multi-universe partial set cover on the device
// (catch/utils/set_cover.py:147-615 with use_intervalsets=True and cost == 1).
//
// State in HBM: one bit per base of the group's concatenated genomes ("still
// uncovered and part of the universe"), the cover rows as CSR (set -> (set,
// universe) segments -> rows), and per-universe counters.  With cost == 1 the
// reference's ratio 1.0/gain orders sets exactly like the integer gain
//   gain(s) = sum_u min(left[u], |s_u ∩ U_u|)
// (float 1.0/n is injective for n < 2^53), ties go to the smallest set id
// (iteration order of the reference's set of dense ids), ranks gate which
// sets may be considered (set_cover.py:497-526).  Each greedy pick is two
// launches, stream-ordered with no host round trip: gain + argmax (one
// wavefront per set, masked popcount over the bitmap words each row touches,
// atomicMax of a packed (gain, ~id) key) and apply (clear the winner's bits,
// update the per-universe counters).  With several GPUs each rank evaluates
// the sets s % nranks == rank and the winner is agreed by one RCCL
// all-reduce(MAX) of the 64-bit key between the two launches.
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
};

#define ID_BITS 24
#define ID_MASK 0xFFFFFFu

__device__ __forceinline__ u32 range_popcount(const u64 *__restrict__ bm, u32 s, u32 e) {
    // bits [s, e), e > s
    u32 w0 = s >> 6, w1 = (e - 1) >> 6;
    u64 m0 = ~0ull << (s & 63);
    u64 m1 = ~0ull >> (63 - ((e - 1) & 63));
    if (w0 == w1) return (u32)__popcll(bm[w0] & m0 & m1);
    u32 c = (u32)__popcll(bm[w0] & m0);
    for (u32 w = w0 + 1; w < w1; ++w) c += (u32)__popcll(bm[w]);
    return c + (u32)__popcll(bm[w1] & m1);
}

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
seg_flag_kernel(const i32 *__restrict__ row_set, const i32 *__restrict__ row_univ, u32 nrows,
                u32 *__restrict__ flag) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    flag[r] = (r == 0 || row_set[r] != row_set[r - 1] || row_univ[r] != row_univ[r - 1]) ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
seg_fill_kernel(const u32 *__restrict__ flag, const u32 *__restrict__ idx, const i32 *__restrict__ row_univ,
                u32 nrows, u32 nseg, u32 *__restrict__ seg_row, u32 *__restrict__ seg_univ) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) seg_row[nseg] = nrows;
    if (r >= nrows || !flag[r]) return;
    seg_row[idx[r]] = r;
    seg_univ[idx[r]] = (u32)row_univ[r];
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

// one wavefront per candidate set
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

// single workgroup: take the winner, remove its elements from the universes
// (set_cover.py:528-550)
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
        // nothing of this rank still covers anything: next rank (set_cover.py:522-526)
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
        u32 n = __hip_atomic_load(&usize[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        if (__hip_atomic_load(&st->n_need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) st->done = 1;
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

    DevBuf<u32> set_ptr, flag, idx, tmp, seg_row, seg_univ, set_seg_ptr, usize, can, left, rank, picked, picks;
    DevBuf<unsigned long long> bm;
    DevBuf<double> d_p;
    DevBuf<GreedyState> st;
    const size_t nwords = (size_t)(R->total / 64 + 2);
    TRY(set_ptr.alloc(nsets + 1));
    TRY(flag.alloc(nrows));
    TRY(idx.alloc(nrows));
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
    hipLaunchKernelGGL(seg_flag_kernel, dim3(rb), dim3(256), 0, s, R->set_id.p, R->univ.p, nrows, flag.p);
    TRY(chip_exclusive_scan_u32(ctx, flag.p, idx.p, nrows, tmp));
    HIP_TRY(hipMemcpyAsync(ctx->h_pin, idx.p + (nrows - 1), sizeof(u32), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync((u32 *)ctx->h_pin + 1, flag.p + (nrows - 1), sizeof(u32), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const u32 nseg = ((volatile u32 *)ctx->h_pin)[0] + ((volatile u32 *)ctx->h_pin)[1];
    TRY(seg_row.alloc(nseg + 1));
    TRY(seg_univ.alloc(nseg + 1));
    hipLaunchKernelGGL(seg_fill_kernel, dim3(rb), dim3(256), 0, s, flag.p, idx.p, R->univ.p, nrows, nseg,
                       seg_row.p, seg_univ.p);
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

    const u32 nranks = (u32)ctx->nranks, myrank = (u32)ctx->rank;
    const u32 my_sets = (u32)div_up((i64)nsets, nranks);
    const unsigned gb = (unsigned)div_up(my_sets, 4);
    const int BATCH = 64;
    const i64 max_iters = (i64)nsets + nrank + 2;
    i64 issued = 0;
    int rc = 0;
    for (;;) {
        for (int b = 0; b < BATCH; ++b) {
            hipLaunchKernelGGL(gain_kernel, dim3(gb), dim3(256), 0, s, (const u64 *)bm.p, R->gs.p, R->ge.p,
                               seg_row.p, seg_univ.p, set_seg_ptr.p, left.p, rank.p, picked.p, nsets, nranks,
                               myrank, st.p);
            if (ctx->comm) {
                ncclResult_t r = ncclAllReduce(&st.p->best_key, &st.p->best_key, 1, ncclUint64, ncclMax,
                                               (ncclComm_t)ctx->comm, s);
                if (r != ncclSuccess) { chip_set_error("ncclAllReduce: %s", ncclGetErrorString(r)); return CATCHHIP_ECOMM; }
            }
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
    if (rc) return rc;
    if (h_st.done == 2) {
        chip_set_error("setcover: ranks exhausted while coverage is still required");
        return CATCHHIP_ERANK;
    }
    std::vector<u32> h_picks(h_st.npicks);
    if (h_st.npicks) {
        HIP_TRY(hipMemcpyAsync(h_picks.data(), picks.p, sizeof(u32) * h_st.npicks, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    for (u32 i = 0; i < h_st.npicks; ++i) out_ids[i] = h_picks[i];
    *n_out = h_st.npicks;
    return 0;
}
