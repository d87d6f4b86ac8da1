// Property checks of a set-cover solution, at any scale, by kernels that share nothing with the solvers
// (setcover*.inc): given the row table of an instance and the picks IN THE ORDER they were made,
//   (1) every pick covered something new when it was taken -- the greedy of catch/utils/set_cover.py:448-550 only
//       ever picks a set with a positive gain -- and
//   (2) every universe ends up covered to its requirement: |covered_u| >= |U_u| - int(|U_u| - p_u |U_u|)
//       (set_cover.py:362-373), U_u = the union of all rows of universe u (:302-320).
// Method: coverage inside one universe depends only on the rows of that universe, so the picks are replayed per
// universe: the rows of the picked sets are keyed (universe, position of their set in the pick order), radix-sorted,
// and ONE THREAD PER UNIVERSE walks its rows in pick order over a bitmap of covered positions, marking a pick as
// soon as one of its rows adds a position.  No gains, no owner words, no rounds: nothing of the frontier solver.
#include "internal.h"

#define CC_NONE 0xffffffffu

__global__ void __launch_bounds__(256)
cc_rank_kernel(const i64 *__restrict__ picks, u32 npicks, u32 nsets, u32 *__restrict__ prank, u32 *__restrict__ bad) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npicks) return;
    const i64 s = picks[i];
    if (s < 0 || s >= (i64)nsets) { atomicAdd(&bad[2], 1u); return; }
    if (atomicCAS(&prank[s], CC_NONE, i) != CC_NONE) atomicAdd(&bad[2], 1u);   // picked twice
}

__device__ __forceinline__ void cc_row_words(u32 gs, u32 ge, u32 *w0, u32 *w1, unsigned long long *m0, unsigned long long *m1) {
    *w0 = gs >> 6; *w1 = (ge - 1) >> 6;
    *m0 = ~0ull << (gs & 63); *m1 = ~0ull >> (63 - ((ge - 1) & 63));
}

// universe bitmap (every row) + flag of the rows of picked sets
__global__ void __launch_bounds__(256)
cc_mark_kernel(const i32 *__restrict__ set_id, const u32 *__restrict__ gs, const u32 *__restrict__ ge, u32 nrows,
               const u32 *__restrict__ prank, u32 nsets, unsigned long long *__restrict__ U, u32 *__restrict__ flag,
               u32 *__restrict__ bad) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const i32 sid = set_id[r];
    // (prank has nsets + 1 entries: a caller whose num_sets is smaller than the rows' largest set id gets
    // bad_pick_ids, not a read out of bounds -- ADVICE round 4)
    if (sid < 0 || (u32)sid >= nsets) { atomicAdd(&bad[2], 1u); flag[r] = 0u; }
    else flag[r] = prank[sid] != CC_NONE ? 1u : 0u;
    if (ge[r] <= gs[r]) return;
    u32 w0, w1; unsigned long long m0, m1;
    cc_row_words(gs[r], ge[r], &w0, &w1, &m0, &m1);
    for (u32 w = w0; w <= w1; ++w) {
        unsigned long long m = ~0ull;
        if (w == w0) m &= m0;
        if (w == w1) m &= m1;
        if ((U[w] & m) != m) atomicOr(&U[w], m);
    }
}

__global__ void __launch_bounds__(256)
cc_keys_kernel(const i32 *__restrict__ set_id, const i32 *__restrict__ univ, u32 nrows, const u32 *__restrict__ prank,
               const u32 *__restrict__ flag, const u32 *__restrict__ pos, u64 *__restrict__ keys, u32 *__restrict__ vals) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows || !flag[r]) return;
    keys[pos[r]] = ((u64)(u32)univ[r] << 32) | prank[set_id[r]];
    vals[pos[r]] = r;
}

// one thread per universe: its picked rows in pick order
__global__ void __launch_bounds__(64)
cc_replay_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ vals, u32 n, const u32 *__restrict__ gs,
                 const u32 *__restrict__ ge, u32 nuniv, unsigned long long *__restrict__ C, u32 *__restrict__ gained,
                 unsigned long long *__restrict__ cov, const unsigned long long *__restrict__ need) {
    const u32 u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nuniv) return;
    // the greedy's gain of a set in universe u is min(left[u], fresh positions) (set_cover.py:531-550): once the
    // universe has met its requirement a pick gains nothing there, whatever it still covers
    const unsigned long long need_u = need[u];
    u32 lo = 0, hi = n;                         // first key of universe u
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if ((u32)(keys[mid] >> 32) < u) lo = mid + 1; else hi = mid; }
    unsigned long long total = 0;
    for (u32 i = lo; i < n && (u32)(keys[i] >> 32) == u; ++i) {
        const u32 r = vals[i];
        if (ge[r] <= gs[r]) continue;
        u32 w0, w1; unsigned long long m0, m1;
        cc_row_words(gs[r], ge[r], &w0, &w1, &m0, &m1);
        u32 fresh = 0;
        for (u32 w = w0; w <= w1; ++w) {
            unsigned long long m = ~0ull;
            if (w == w0) m &= m0;
            if (w == w1) m &= m1;
            // (a word on the border of two universes is shared: only this thread ever touches these bits of it)
            const unsigned long long add = m & ~C[w];
            if (add) { fresh += (u32)__popcll(add); atomicOr(&C[w], add); }
        }
        if (fresh) { if (total < need_u) gained[(u32)keys[i]] = 1u; total += fresh; }
    }
    cov[u] = total;
}

// |U_u| and the requirement of a universe; int(n - p n) with IEEE doubles, no fused multiply-add (set_cover.py:362-373)
__global__ void __launch_bounds__(64)
cc_need_kernel(const unsigned long long *__restrict__ U, const u32 *__restrict__ genome_off, u32 nuniv,
               const double *__restrict__ p, unsigned long long *__restrict__ usize_out,
               unsigned long long *__restrict__ need_out) {
    const u32 u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nuniv) return;
    const u32 g0 = genome_off[u], g1 = genome_off[u + 1];
    unsigned long long usize = 0;
    if (g1 > g0) {
        const u32 w0 = g0 >> 6, w1 = (g1 - 1) >> 6;
        for (u32 w = w0; w <= w1; ++w) {
            unsigned long long m = ~0ull;
            if (w == w0) m &= ~0ull << (g0 & 63);
            if (w == w1) m &= ~0ull >> (63 - ((g1 - 1) & 63));
            usize += (unsigned long long)__popcll(U[w] & m);
        }
    }
    const double n = (double)usize, pu = p ? p[u] : 1.0;
    long long can = (long long)__dsub_rn(n, __dmul_rn(pu, n));
    if (can < 0) can = 0;
    if (can > (long long)usize) can = (long long)usize;
    usize_out[u] = usize;
    need_out[u] = usize - (unsigned long long)can;
}

__global__ void __launch_bounds__(64)
cc_universe_kernel(const unsigned long long *__restrict__ usize, const unsigned long long *__restrict__ need, u32 nuniv,
                   const unsigned long long *__restrict__ cov, u32 *__restrict__ bad, unsigned long long *__restrict__ sums) {
    const u32 u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nuniv) return;
    if (cov[u] < need[u]) atomicAdd(&bad[1], 1u);
    atomicAdd(&sums[0], usize[u]);
    atomicAdd(&sums[1], cov[u]);
}

__global__ void __launch_bounds__(256)
cc_picks_kernel(const u32 *__restrict__ gained, u32 npicks, u32 *__restrict__ bad) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool b = i < npicks && !gained[i];
    const unsigned long long m = __ballot(b);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&bad[0], (u32)__popcll(m));
}

extern "C" int catchhip_rows_cover_check(catchhip_ctx *ctx, const catchhip_rows *R, i64 num_sets, const i64 *picks,
                                         i64 npicks, const double *universe_p, i64 *out5) {
    ARG_CHECK(ctx && R && out5 && num_sets >= 0 && npicks >= 0 && (npicks == 0 || picks));
    ARG_CHECK(R->ctx == ctx && !R->deferred && num_sets < ((i64)1 << 32) && npicks < ((i64)1 << 32));
    HIP_TRY(hipSetDevice(ctx->device));
    PoolScope pool_scope(ctx);
    hipStream_t s = ctx->stream;
    const u32 nrows = (u32)R->n, nuniv = (u32)R->ngenomes, nsets = (u32)num_sets, np = (u32)npicks;
    const size_t nwords = (size_t)(R->total / 64 + 2);
    for (int i = 0; i < 5; ++i) out5[i] = 0;
    DevBuf<u32> prank, flag, pos, tmp, gained, bad, vals, vals_alt;
    DevBuf<u64> keys, keys_alt;
    DevBuf<unsigned long long> U, C, cov, sums, usize, need;
    DevBuf<i64> d_picks;
    DevBuf<double> d_p;
    TRY(prank.alloc((size_t)nsets + 1));
    TRY(flag.alloc((size_t)nrows + 1));
    TRY(pos.alloc((size_t)nrows + 1));
    TRY(gained.alloc((size_t)np + 1));
    TRY(bad.alloc(4));
    TRY(U.alloc(nwords));
    TRY(C.alloc(nwords));
    TRY(cov.alloc((size_t)nuniv + 1));
    TRY(sums.alloc(2));
    TRY(usize.alloc((size_t)nuniv + 1));
    TRY(need.alloc((size_t)nuniv + 1));
    TRY(d_picks.alloc((size_t)np + 1));
    HIP_TRY(hipMemsetAsync(prank.p, 0xff, sizeof(u32) * ((size_t)nsets + 1), s));
    HIP_TRY(hipMemsetAsync(gained.p, 0, sizeof(u32) * ((size_t)np + 1), s));
    HIP_TRY(hipMemsetAsync(bad.p, 0, sizeof(u32) * 4, s));
    HIP_TRY(hipMemsetAsync(U.p, 0, sizeof(unsigned long long) * nwords, s));
    HIP_TRY(hipMemsetAsync(C.p, 0, sizeof(unsigned long long) * nwords, s));
    HIP_TRY(hipMemsetAsync(cov.p, 0, sizeof(unsigned long long) * ((size_t)nuniv + 1), s));
    HIP_TRY(hipMemsetAsync(sums.p, 0, sizeof(unsigned long long) * 2, s));
    if (np) HIP_TRY(hipMemcpyAsync(d_picks.p, picks, sizeof(i64) * np, hipMemcpyHostToDevice, s));
    if (universe_p && nuniv) {
        TRY(d_p.alloc(nuniv));
        HIP_TRY(hipMemcpyAsync(d_p.p, universe_p, sizeof(double) * nuniv, hipMemcpyHostToDevice, s));
    }
    if (np) hipLaunchKernelGGL(cc_rank_kernel, dim3((unsigned)div_up(np, 256)), dim3(256), 0, s, (const i64 *)d_picks.p, np,
                               nsets, prank.p, bad.p);
    u32 nkeys = 0;
    if (nrows) {
        hipLaunchKernelGGL(cc_mark_kernel, dim3((unsigned)div_up(nrows, 256)), dim3(256), 0, s, (const i32 *)R->set_id.p,
                           (const u32 *)R->gs.p, (const u32 *)R->ge.p, nrows, (const u32 *)prank.p, nsets, U.p, flag.p, bad.p);
        HIP_TRY(hipMemsetAsync(flag.p + nrows, 0, sizeof(u32), s));
        TRY(chip_exclusive_scan_u32(ctx, flag.p, pos.p, (i64)nrows + 1, tmp));
        HIP_TRY(hipMemcpyAsync(ctx->h_pin, pos.p + nrows, sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        nkeys = *(volatile u32 *)ctx->h_pin;
    }
    TRY(keys.alloc(std::max<u32>(nkeys, 1)));
    TRY(vals.alloc(std::max<u32>(nkeys, 1)));
    if (nkeys) {
        hipLaunchKernelGGL(cc_keys_kernel, dim3((unsigned)div_up(nrows, 256)), dim3(256), 0, s, (const i32 *)R->set_id.p,
                           (const i32 *)R->univ.p, nrows, (const u32 *)prank.p, (const u32 *)flag.p, (const u32 *)pos.p,
                           keys.p, vals.p);
        TRY(chip_radix_sort_pairs(ctx, keys, keys_alt, vals, vals_alt, (i64)nkeys, 32 + ceil_log2_u64((u64)nuniv + 1)));
    }
    if (nuniv) {
        hipLaunchKernelGGL(cc_need_kernel, dim3((unsigned)div_up(nuniv, 64)), dim3(64), 0, s,
                           (const unsigned long long *)U.p, (const u32 *)R->genome_off.p, nuniv,
                           universe_p ? (const double *)d_p.p : (const double *)nullptr, usize.p, need.p);
        hipLaunchKernelGGL(cc_replay_kernel, dim3((unsigned)div_up(nuniv, 64)), dim3(64), 0, s, (const u64 *)keys.p,
                           (const u32 *)vals.p, nkeys, (const u32 *)R->gs.p, (const u32 *)R->ge.p, nuniv, C.p, gained.p, cov.p,
                           (const unsigned long long *)need.p);
        hipLaunchKernelGGL(cc_universe_kernel, dim3((unsigned)div_up(nuniv, 64)), dim3(64), 0, s,
                           (const unsigned long long *)usize.p, (const unsigned long long *)need.p, nuniv,
                           (const unsigned long long *)cov.p, bad.p, sums.p);
    }
    if (np) hipLaunchKernelGGL(cc_picks_kernel, dim3((unsigned)div_up(np, 256)), dim3(256), 0, s, (const u32 *)gained.p, np, bad.p);
    HIP_TRY(hipGetLastError());
    u32 hb[4];
    unsigned long long hs[2];
    HIP_TRY(hipMemcpyAsync(hb, bad.p, sizeof(hb), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hs, sums.p, sizeof(hs), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    out5[0] = hb[0];            // picks that covered nothing new at their turn
    out5[1] = hb[1];            // universes short of their requirement
    out5[2] = hb[2];            // pick ids out of range or repeated
    out5[3] = (i64)hs[0];       // bases in the universes
    out5[4] = (i64)hs[1];       // of them covered by the picks
    return 0;
}
