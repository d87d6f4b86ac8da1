// Candidate probes and their exact de-duplication on the device.
//
// Replaces, for the usual filter list [DuplicateFilter, SetCoverFilter]:
//   * candidate_probes.make_candidate_probes_from_sequences
//     (catch/filter/candidate_probes.py:21-182): per sequence the windows of
//     probe_length every probe_stride bases, a last window flush with the end
//     when the length is not a multiple of the stride, windows holding two or
//     more consecutive N dropped (:63-70), and for every maximal run of >= 2 N
//     the windows just left and right of it (:104-122) -- in exactly that
//     order, sequences in genome order;
//   * DuplicateFilter (catch/filter/duplicate_filter.py:16-26): first
//     occurrences, order kept -- the position in that list is the candidate's
//     set id, which the greedy solver's tie-break depends on.
// The unique candidates never become strings: catchhip_probes_from_candidates
// gathers them from the targets' characters into a probes object, and only the
// selected ones are looked up again by the host (catchhip_candidates_fetch).
//
// Kernels (all streaming integer work, HBM-bound):
//   cand_flags_kernel    per base: "an NN pair starts here", "a run of >= 2 N starts here"
//   (two prefix sums)    pairs before a position (window validity in O(1)), runs before a position
//   cand_seq_kernel      per sequence: windows, runs -> slots
//   cand_regular_kernel  one thread per stride window (+ the tail window)
//   cand_flank_kernel    one thread per N run: its two flanking windows
//   (compaction)         valid slots -> candidate list in reference order
//   cand_hash_kernel     64-bit hash of every candidate's characters
//   (radix sort)         equal hashes adjacent, candidate order kept inside a run
//   cand_dup_kernel      compare with the run's first candidate byte by byte;
//                        a hash collision of different strings (never seen)
//                        falls back to a walk over the run
//   (compaction)         first occurrences in candidate order
#include <algorithm>

#include "internal.h"

struct catchhip_candidates {
    catchhip_ctx *ctx = nullptr;
    const catchhip_targets *T = nullptr;   // borrowed: must outlive this object
    i32 L = 0;
    i64 ncand = 0, nuniq = 0;
    DevBuf<u32> upos;   // global start of every unique candidate, first-occurrence order
    DevBuf<u32> mult;   // how many candidates equal each unique one (valid until a near-duplicate filter ran)
    bool grouped = false;   // the targets carry groups: duplicates are only removed inside a group
    i32 ngroups = 0;
    DevBuf<u32> ugrp;   // group of every unique candidate (non-decreasing)
    bool filtered = false;   // a near-duplicate filter replaced the list (multiplicity order, kept ones only)
};

#define CAND_NONE 0xffffffffu

__device__ __forceinline__ u32 cand_find_segment(const u32 *__restrict__ off, u32 n, u32 x) {
    u32 lo = 0, hi = n;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (off[mid] <= x) lo = mid; else hi = mid;
    }
    while (lo + 1 < n && off[lo + 1] <= x) ++lo;
    return lo;
}

__global__ void __launch_bounds__(256)
cand_flags_kernel(const u8 *__restrict__ bytes, const u32 *__restrict__ seq_off, u32 nseq, u32 total,
                  u32 *__restrict__ pair, u32 *__restrict__ rstart) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > total) return;
    u32 pr = 0, rs = 0;
    if (g < total && bytes[g] == 'N') {
        const u32 sq = cand_find_segment(seq_off, nseq, g);
        const u32 lo = seq_off[sq], hi = seq_off[sq + 1];
        pr = (g + 1 < hi && bytes[g + 1] == 'N') ? 1u : 0u;
        rs = (pr && (g == lo || bytes[g - 1] != 'N')) ? 1u : 0u;
    }
    pair[g] = pr;      // entry `total` is the sentinel of the prefix sums
    rstart[g] = rs;
}

// a window [s, s+L) holds two consecutive N iff an NN pair starts in [s, s+L-2]
__device__ __forceinline__ bool cand_window_ok(const u32 *__restrict__ pairs_before, u32 s, u32 L) {
    return pairs_before[s + L - 1] == pairs_before[s];
}

__global__ void __launch_bounds__(256)
cand_seq_kernel(const u32 *__restrict__ seq_off, u32 nseq, u32 L, u32 stride, i64 skip_len,
                const u32 *__restrict__ runs_before, u32 *__restrict__ nwin, u32 *__restrict__ nslot) {
    const u32 q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nseq) return;
    u32 w = 0, sl = 0;
    if (q < nseq) {
        const u32 lo = seq_off[q], hi = seq_off[q + 1], n = hi - lo;
        if (!(skip_len >= 0 && (i64)n <= skip_len) && n >= L) {
            w = (n - L) / stride + 1 + 1;                       // stride windows + the tail slot
            sl = w + 2 * (runs_before[hi] - runs_before[lo]);   // + two flanks per run
        }
    }
    nwin[q] = w;
    nslot[q] = sl;
}

__global__ void __launch_bounds__(256)
cand_regular_kernel(const u32 *__restrict__ seq_off, u32 nseq, u32 L, u32 stride, const u32 *__restrict__ win_off,
                    const u32 *__restrict__ slot_off, const u32 *__restrict__ pairs_before, u32 nwin_total,
                    u32 *__restrict__ slots) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nwin_total) return;
    const u32 q = cand_find_segment(win_off, nseq, t);
    const u32 w = t - win_off[q], nw = win_off[q + 1] - win_off[q];
    const u32 lo = seq_off[q], hi = seq_off[q + 1], n = hi - lo;
    u32 s = CAND_NONE;
    if (w + 1 < nw) s = lo + w * stride;
    else if (n % stride != 0) s = hi - L;                       // candidate_probes.py:99-102
    if (s != CAND_NONE && !cand_window_ok(pairs_before, s, L)) s = CAND_NONE;
    slots[slot_off[q] + w] = s;
}

__global__ void __launch_bounds__(256)
cand_flank_kernel(const u8 *__restrict__ bytes, const u32 *__restrict__ seq_off, u32 nseq, u32 total, u32 L,
                  const u32 *__restrict__ rstart, const u32 *__restrict__ runs_before,
                  const u32 *__restrict__ win_off, const u32 *__restrict__ slot_off,
                  const u32 *__restrict__ pairs_before, u32 *__restrict__ slots) {
    const u32 a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= total || !rstart[a]) return;
    const u32 q = cand_find_segment(seq_off, nseq, a);
    const u32 nw = win_off[q + 1] - win_off[q];
    if (nw == 0) return;                                        // skipped sequence
    const u32 lo = seq_off[q], hi = seq_off[q + 1];
    u32 b = a;
    while (b < hi && bytes[b] == 'N') ++b;
    const u32 r = runs_before[a] - runs_before[lo];
    const u32 at = slot_off[q] + nw + 2 * r;
    u32 left = CAND_NONE, right = CAND_NONE;
    if (a - lo >= L && cand_window_ok(pairs_before, a - L, L)) left = a - L;      // :108-114
    if (b + L <= hi && cand_window_ok(pairs_before, b, L)) right = b;              // :115-121
    slots[at] = left;
    slots[at + 1] = right;
}

__global__ void __launch_bounds__(256)
cand_valid_kernel(const u32 *__restrict__ slots, u32 n, u32 *__restrict__ flag) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= n) flag[t] = (t < n && slots[t] != CAND_NONE) ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
cand_compact_kernel(const u32 *__restrict__ src, const u32 *__restrict__ flag, const u32 *__restrict__ at, u32 n,
                    u32 *__restrict__ dst) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && flag[t]) dst[at[t]] = src[t];
}

__global__ void __launch_bounds__(256)
cand_hash_kernel(const u8 *__restrict__ bytes, const u32 *__restrict__ cpos, u32 n, u32 L, u64 *__restrict__ keys,
                 u32 *__restrict__ vals, const u32 *__restrict__ seq_off, u32 nseq, const i32 *__restrict__ seq_group,
                 u32 *__restrict__ cgrp, u64 hash_mask) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const u8 *p = bytes + cpos[c];
    u64 h = 0x9e3779b97f4a7c15ull ^ (u64)L;
    if (seq_group) {   // independent groups: equal windows of different groups are different candidates
        const u32 g = (u32)seq_group[cand_find_segment(seq_off, nseq, cpos[c])];
        cgrp[c] = g;
        h = (h ^ (u64)g) * 0xff51afd7ed558ccdull;
    }
    u32 j = 0;
    for (; j + 8 <= L; j += 8) {
        u64 w = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) w |= (u64)p[j + t] << (8 * t);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 29;
    }
    u64 w = 0;
    for (u32 t = 0; j + t < L; ++t) w |= (u64)p[j + t] << (8 * t);
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 32;
    keys[c] = h & hash_mask;   // all ones, except in tests that provoke collisions
    vals[c] = c;
}

__global__ void __launch_bounds__(256)
cand_runstart_kernel(const u64 *__restrict__ keys, u32 n, u32 *__restrict__ flag) {
    const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x <= n) flag[x] = (x < n && (x == 0 || keys[x] != keys[x - 1])) ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
cand_runhead_kernel(const u32 *__restrict__ flag, const u32 *__restrict__ runid, u32 n, u32 *__restrict__ head) {
    const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n && flag[x]) head[runid[x]] = x;
}

__device__ __forceinline__ bool cand_same(const u8 *__restrict__ bytes, u32 a, u32 b, u32 L) {
    for (u32 j = 0; j < L; ++j)
        if (bytes[a + j] != bytes[b + j]) return false;
    return true;
}

// keep[c] = 1 iff no earlier candidate has the same characters.  Inside a run of
// equal hashes the stable sort keeps candidates in ascending order, so the run's
// first element is the earliest.
__global__ void __launch_bounds__(256)
cand_dup_kernel(const u8 *__restrict__ bytes, const u32 *__restrict__ cpos, const u32 *__restrict__ vals,
                const u32 *__restrict__ flag, const u32 *__restrict__ runid, const u32 *__restrict__ head, u32 n,
                u32 L, u32 *__restrict__ keep, u32 *__restrict__ mult, const u32 *__restrict__ cgrp) {
    const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x > n) return;
    if (x == n) { keep[n] = 0; return; }
    const u32 c = vals[x];
    // runid is an exclusive sum of the start flags: a run's own start counts for its later elements only
    const u32 h = flag[x] ? x : head[runid[x] - 1];
    u32 k = 1, rep = c;   // rep: the earliest candidate with these characters
    if (x != h) {
        const u32 g = cgrp ? cgrp[c] : 0u;
        if ((!cgrp || cgrp[vals[h]] == g) && cand_same(bytes, cpos[c], cpos[vals[h]], L)) { k = 0; rep = vals[h]; }
        else
            for (u32 y = h + 1; y < x; ++y)   // different strings under one hash: look at the others, earliest first
                if ((!cgrp || cgrp[vals[y]] == g) && cand_same(bytes, cpos[c], cpos[vals[y]], L)) { k = 0; rep = vals[y]; break; }
    }
    keep[c] = k;
    atomicAdd(&mult[rep], 1u);   // multiplicity of the unique candidate (the near-duplicate filters' priority)
}

__global__ void __launch_bounds__(256)
cand_gather_kernel(const u8 *__restrict__ tbytes, const u32 *__restrict__ upos, u32 n, u32 L, u8 *__restrict__ out,
                   u32 *__restrict__ probe_off, i32 *__restrict__ set_id, u32 *__restrict__ bucket_of,
                   i32 *__restrict__ bucket_set) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 nb = (u64)n * L;
    if (t < nb) {
        const u32 i = (u32)(t / L), j = (u32)(t - (u64)i * L);
        out[t] = tbytes[upos[i] + j];
    }
    if (t <= n) {
        probe_off[t] = (u32)(t * L);
        if (t < n) { set_id[t] = (i32)t; bucket_of[t] = (u32)t; bucket_set[t] = (i32)t; }
    }
}

__global__ void __launch_bounds__(256)
cand_pigeon_kernel(u32 n, u32 nanch, u32 k, i32 *__restrict__ ent_probe, i32 *__restrict__ ent_pos,
                   u32 *__restrict__ sent_probe, u32 *__restrict__ sent_pos, u32 *__restrict__ ent_ptr) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 ne = (u64)n * nanch;
    if (t < ne) {
        const u32 p = (u32)(t / nanch), a = (u32)(t - (u64)p * nanch) * k;
        ent_probe[t] = (i32)p; ent_pos[t] = (i32)a;
        sent_probe[t] = p; sent_pos[t] = a;
    }
    if (t <= n) ent_ptr[t] = (u32)(t * nanch);
}

static int cand_scan(catchhip_ctx *ctx, DevBuf<u32> &flag, DevBuf<u32> &out, i64 n_plus_1, DevBuf<u32> &tmp) {
    TRY(out.reserve((size_t)n_plus_1));
    return chip_exclusive_scan_u32(ctx, flag.p, out.p, n_plus_1, tmp);
}

static int cand_read_u32(catchhip_ctx *ctx, const u32 *d, u32 *out) {
    HIP_TRY(hipMemcpyAsync(ctx->h_pin, d, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = *(volatile u32 *)ctx->h_pin;
    return 0;
}

static int candidates_build(catchhip_ctx *ctx, const catchhip_targets *T, u32 L, u32 stride, i64 skip_len,
                            catchhip_candidates *C) {
    hipStream_t s = ctx->stream;
    const u32 total = (u32)T->total, nseq = (u32)T->nseq;
    DevBuf<u32> pair, rstart, pairs_before, runs_before, nwin, nslot, win_off, slot_off, slots, flag, at, cpos, tmp;
    PhaseTimer tm(ctx, PHASE_NDF);
    TRY(pair.alloc((size_t)total + 1));
    TRY(rstart.alloc((size_t)total + 1));
    hipLaunchKernelGGL(cand_flags_kernel, dim3((unsigned)div_up((i64)total + 1, 256)), dim3(256), 0, s,
                       (const u8 *)T->bytes.p, (const u32 *)T->seq_off.p, nseq, total, pair.p, rstart.p);
    TRY(cand_scan(ctx, pair, pairs_before, (i64)total + 1, tmp));
    TRY(cand_scan(ctx, rstart, runs_before, (i64)total + 1, tmp));
    TRY(nwin.alloc((size_t)nseq + 1));
    TRY(nslot.alloc((size_t)nseq + 1));
    hipLaunchKernelGGL(cand_seq_kernel, dim3((unsigned)div_up((i64)nseq + 1, 256)), dim3(256), 0, s,
                       (const u32 *)T->seq_off.p, nseq, L, stride, skip_len, (const u32 *)runs_before.p, nwin.p, nslot.p);
    TRY(cand_scan(ctx, nwin, win_off, (i64)nseq + 1, tmp));
    TRY(cand_scan(ctx, nslot, slot_off, (i64)nseq + 1, tmp));
    u32 nwin_total = 0, nslot_total = 0;
    TRY(cand_read_u32(ctx, win_off.p + nseq, &nwin_total));
    TRY(cand_read_u32(ctx, slot_off.p + nseq, &nslot_total));
    tm.launch(12);
    C->ncand = 0;
    C->nuniq = 0;
    if (nslot_total == 0) { tm.finish(); return 0; }
    TRY(slots.alloc((size_t)nslot_total + 1));
    HIP_TRY(hipMemsetAsync(slots.p, 0xff, sizeof(u32) * ((size_t)nslot_total + 1), s));
    hipLaunchKernelGGL(cand_regular_kernel, dim3((unsigned)div_up((i64)nwin_total, 256)), dim3(256), 0, s,
                       (const u32 *)T->seq_off.p, nseq, L, stride, (const u32 *)win_off.p, (const u32 *)slot_off.p,
                       (const u32 *)pairs_before.p, nwin_total, slots.p);
    hipLaunchKernelGGL(cand_flank_kernel, dim3((unsigned)div_up((i64)total, 256)), dim3(256), 0, s,
                       (const u8 *)T->bytes.p, (const u32 *)T->seq_off.p, nseq, total, L, (const u32 *)rstart.p,
                       (const u32 *)runs_before.p, (const u32 *)win_off.p, (const u32 *)slot_off.p,
                       (const u32 *)pairs_before.p, slots.p);
    TRY(flag.alloc((size_t)nslot_total + 1));
    hipLaunchKernelGGL(cand_valid_kernel, dim3((unsigned)div_up((i64)nslot_total + 1, 256)), dim3(256), 0, s,
                       (const u32 *)slots.p, nslot_total, flag.p);
    TRY(cand_scan(ctx, flag, at, (i64)nslot_total + 1, tmp));
    u32 ncand = 0;
    TRY(cand_read_u32(ctx, at.p + nslot_total, &ncand));
    tm.launch(6);
    C->ncand = ncand;
    if (ncand == 0) { tm.finish(); return 0; }
    TRY(cpos.alloc(ncand));
    hipLaunchKernelGGL(cand_compact_kernel, dim3((unsigned)div_up((i64)nslot_total, 256)), dim3(256), 0, s,
                       (const u32 *)slots.p, (const u32 *)flag.p, (const u32 *)at.p, nslot_total, cpos.p);
    // the big per-base arrays are not needed any more
    pair.release(); rstart.release(); pairs_before.release(); runs_before.release(); slots.release();
    // exact de-duplication
    DevBuf<u64> keys, keys_alt;
    DevBuf<u32> vals, vals_alt, rflag, runid, head, keep, kat;
    DevBuf<u32> cgrp;
    // CATCHHIP_CAND_HASH_BITS (tests): keep only that many bits of the hash, so that
    // different windows collide and the byte-wise resolution is exercised
    u64 hash_mask = ~0ull;
    if (const char *e = chip_test_env("CATCHHIP_CAND_HASH_BITS")) {
        const int b = atoi(e);
        if (b >= 0 && b < 64) hash_mask = ((u64)1 << b) - 1;
    }
    TRY(keys.alloc(ncand));
    TRY(vals.alloc(ncand));
    if (C->grouped) TRY(cgrp.alloc((size_t)ncand + 1));
    const u32 *cg = C->grouped ? (const u32 *)cgrp.p : (const u32 *)nullptr;
    hipLaunchKernelGGL(cand_hash_kernel, dim3((unsigned)div_up((i64)ncand, 256)), dim3(256), 0, s,
                       (const u8 *)T->bytes.p, (const u32 *)cpos.p, ncand, L, keys.p, vals.p,
                       (const u32 *)T->seq_off.p, nseq, C->grouped ? (const i32 *)T->seq_group.p : (const i32 *)nullptr,
                       cgrp.p, hash_mask);
    TRY(chip_radix_sort_pairs(ctx, keys, keys_alt, vals, vals_alt, ncand, 64));
    TRY(rflag.alloc((size_t)ncand + 1));
    hipLaunchKernelGGL(cand_runstart_kernel, dim3((unsigned)div_up((i64)ncand + 1, 256)), dim3(256), 0, s,
                       (const u64 *)keys.p, ncand, rflag.p);
    TRY(cand_scan(ctx, rflag, runid, (i64)ncand + 1, tmp));
    TRY(head.alloc((size_t)ncand + 1));
    // runid (exclusive) of a run start = index of its run
    hipLaunchKernelGGL(cand_runhead_kernel, dim3((unsigned)div_up((i64)ncand, 256)), dim3(256), 0, s,
                       (const u32 *)rflag.p, (const u32 *)runid.p, ncand, head.p);
    DevBuf<u32> cmult;
    TRY(keep.alloc((size_t)ncand + 1));
    TRY(cmult.alloc((size_t)ncand + 1));
    HIP_TRY(hipMemsetAsync(cmult.p, 0, sizeof(u32) * ((size_t)ncand + 1), s));
    hipLaunchKernelGGL(cand_dup_kernel, dim3((unsigned)div_up((i64)ncand + 1, 256)), dim3(256), 0, s,
                       (const u8 *)T->bytes.p, (const u32 *)cpos.p, (const u32 *)vals.p, (const u32 *)rflag.p,
                       (const u32 *)runid.p, (const u32 *)head.p, ncand, L, keep.p, cmult.p, cg);
    TRY(cand_scan(ctx, keep, kat, (i64)ncand + 1, tmp));
    u32 nuniq = 0;
    TRY(cand_read_u32(ctx, kat.p + ncand, &nuniq));
    C->nuniq = nuniq;
    TRY(C->upos.alloc((size_t)nuniq + 1));
    TRY(C->mult.alloc((size_t)nuniq + 1));
    hipLaunchKernelGGL(cand_compact_kernel, dim3((unsigned)div_up((i64)ncand, 256)), dim3(256), 0, s,
                       (const u32 *)cpos.p, (const u32 *)keep.p, (const u32 *)kat.p, ncand, C->upos.p);
    hipLaunchKernelGGL(cand_compact_kernel, dim3((unsigned)div_up((i64)ncand, 256)), dim3(256), 0, s,
                       (const u32 *)cmult.p, (const u32 *)keep.p, (const u32 *)kat.p, ncand, C->mult.p);
    if (C->grouped) {
        TRY(C->ugrp.alloc((size_t)nuniq + 1));
        hipLaunchKernelGGL(cand_compact_kernel, dim3((unsigned)div_up((i64)ncand, 256)), dim3(256), 0, s,
                           (const u32 *)cgrp.p, (const u32 *)keep.p, (const u32 *)kat.p, ncand, C->ugrp.p);
    }
    tm.launch(41);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    tm.finish();
    return 0;
}

extern "C" int catchhip_candidates_create(catchhip_ctx *ctx, const catchhip_targets *T, i32 probe_length,
                                          i32 probe_stride, i64 seq_length_to_skip, catchhip_candidates **out,
                                          i64 *ncandidates, i64 *nunique) {
    ARG_CHECK(ctx && T && T->ctx == ctx && out && probe_length >= 2 && probe_stride >= 1);
    PoolScope pool_scope(ctx);
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    // candidate_probes.py:37-50: a sequence shorter than the probe is an error
    // (or a probe of its own with --small-seq-min: the host's string path)
    for (i64 q = 0; q < T->nseq; ++q) {
        const i64 n = T->h_seq_off[(size_t)q + 1] - T->h_seq_off[(size_t)q];
        if (seq_length_to_skip >= 0 && n <= seq_length_to_skip) continue;
        if (n < probe_length) {
            chip_set_error("candidates: sequence %lld is shorter than the probe length %d", (long long)q,
                           (int)probe_length);
            return CATCHHIP_EINVAL;
        }
    }
    catchhip_candidates *C = new catchhip_candidates();
    C->ctx = ctx;
    C->T = T;
    C->L = probe_length;
    C->grouped = T->has_groups;
    C->ngroups = T->ngroups_set;
    int rc = T->total ? candidates_build(ctx, T, (u32)probe_length, (u32)probe_stride, seq_length_to_skip, C) : 0;
    if (rc) { delete C; return rc; }
    if (ncandidates) *ncandidates = C->ncand;
    if (nunique) *nunique = C->nuniq;
    *out = C;
    return 0;
}

// see catchhip_targets_rebind (core.hip)
extern "C" int catchhip_candidates_rebind(catchhip_candidates *C, catchhip_ctx *to) {
    ARG_CHECK(C && C->ctx && to && C->ctx->device == to->device);
    HIP_TRY(hipSetDevice(to->device));
    if (C->ctx != to) HIP_TRY(hipStreamSynchronize(C->ctx->stream));
    C->ctx = to;
    return 0;
}

extern "C" void catchhip_candidates_destroy(catchhip_candidates *C) {
    if (!C) return;
    PoolScope pool_scope(C->ctx);
    delete C;
}

extern "C" int catchhip_candidates_fetch(catchhip_ctx *ctx, const catchhip_candidates *C, const i64 *ids, i64 n,
                                         i64 *global_start) {
    ARG_CHECK(ctx && C && C->ctx == ctx && n >= 0);
    if (n == 0) return 0;
    ARG_CHECK(global_start);
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<u32> h((size_t)C->nuniq);
    if (C->nuniq) {
        HIP_TRY(hipMemcpyAsync(h.data(), C->upos.p, sizeof(u32) * (size_t)C->nuniq, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    for (i64 i = 0; i < n; ++i) {
        const i64 id = ids ? ids[i] : i;
        if (id < 0 || id >= C->nuniq) { chip_set_error("candidates_fetch: id out of range"); return CATCHHIP_ERANK; }
        global_start[i] = h[(size_t)id];
    }
    return 0;
}

// ---- near-duplicate filters on the device's candidates ------------------------
__global__ void __launch_bounds__(256)
cand_multkey_kernel(const u32 *__restrict__ mult, const u32 *__restrict__ ugrp, u32 n, u64 *__restrict__ keys,
                    u32 *__restrict__ vals) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // ascending = (group,) multiplicity descending; the sort is stable
    keys[i] = ((u64)(ugrp ? ugrp[i] : 0u) << 32) | (u64)(0xffffffffu - mult[i]);
    vals[i] = i;
}

__global__ void __launch_bounds__(256)
cand_permute_kernel(const u32 *__restrict__ upos, const u32 *__restrict__ order, u32 n, u32 *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = upos[order[i]];
}

__global__ void __launch_bounds__(256)
cand_rows_kernel(const u8 *__restrict__ tbytes, const u32 *__restrict__ pos, u32 n, u32 L, u8 *__restrict__ out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (u64)n * L) return;
    const u32 i = (u32)(t / L), j = (u32)(t - (u64)i * L);
    out[t] = tbytes[pos[i] + j];
}

// the unique candidates in the near-duplicate filters' priority order (multiplicity
// descending, ties in first-occurrence order: near_duplicate_filter.py:60-66) and
// their characters as rows
static int cand_priority_rows(catchhip_ctx *ctx, catchhip_candidates *C, DevBuf<u32> &opos, DevBuf<u8> &rows,
                              DevBuf<u32> &ogrp) {
    hipStream_t s = ctx->stream;
    const u32 n = (u32)C->nuniq, L = (u32)C->L;
    DevBuf<u64> keys, keys_alt;
    DevBuf<u32> vals, vals_alt;
    TRY(keys.alloc(n));
    TRY(vals.alloc(n));
    hipLaunchKernelGGL(cand_multkey_kernel, dim3((unsigned)div_up((i64)n, 256)), dim3(256), 0, s,
                       (const u32 *)C->mult.p, C->grouped ? (const u32 *)C->ugrp.p : (const u32 *)nullptr, n, keys.p,
                       vals.p);
    TRY(chip_radix_sort_pairs(ctx, keys, keys_alt, vals, vals_alt, n, C->grouped ? 64 : 32));
    TRY(opos.alloc((size_t)n + 1));
    hipLaunchKernelGGL(cand_permute_kernel, dim3((unsigned)div_up((i64)n, 256)), dim3(256), 0, s,
                       (const u32 *)C->upos.p, (const u32 *)vals.p, n, opos.p);
    if (C->grouped) {
        TRY(ogrp.alloc((size_t)n + 1));
        hipLaunchKernelGGL(cand_permute_kernel, dim3((unsigned)div_up((i64)n, 256)), dim3(256), 0, s,
                           (const u32 *)C->ugrp.p, (const u32 *)vals.p, n, ogrp.p);
    }
    TRY(rows.alloc((size_t)n * L + 64));
    HIP_TRY(hipMemsetAsync(rows.p + (size_t)n * L, 0, 64, s));
    hipLaunchKernelGGL(cand_rows_kernel, dim3((unsigned)div_up((i64)n * L, 256)), dim3(256), 0, s,
                       (const u8 *)C->T->bytes.p, (const u32 *)opos.p, n, L, rows.p);
    HIP_TRY(hipGetLastError());
    return 0;
}

// hash(seq_str) of every candidate (PYTHONHASHSEED=0 semantics, internal.h)
__global__ void __launch_bounds__(256)
cand_pyhash_kernel(const u8 *__restrict__ tbytes, const u32 *__restrict__ pos, u32 n, u32 L, long long *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = chip_pyhash_seed0(tbytes + pos[i], (int)L);
}

// ---- the iteration order of a CPython set, built on the device --------------------------
// chip_pyset_order (core.hip) inserts one key after the other; here all insertions of one table
// generation run at once.  What sequential insertion computes is the one assignment in which
// every key sits in the first slot of its probe sequence that no EARLIER key holds, so the
// slots hold (priority, index) words, a key claims a slot with atomicMin and whoever it
// displaces walks on from that slot: the fixed point is the sequential table whatever the
// interleaving (slot words only ever decrease, so a slot once lost to a key stays lost).
// A rebuild re-inserts the entries in slot order -- priority = position in the previous
// generation's iteration order -- and the keys added afterwards follow in input order.
#define PYSET_EMPTY (~0ull)

struct PysetProbe {       // setobject.c set_add_entry / set_insert_clean: slot, 9 linear probes, perturbed jump
    u64 base, perturb, mask;
    u32 j;
    __device__ __forceinline__ void start(u64 h, u64 m) { mask = m; perturb = h; base = h & m; j = 0; }
    __device__ __forceinline__ u64 slot() const { return base + j; }
    __device__ __forceinline__ void next() {
        if (j == 0 ? (base + 9 <= mask) : (j < 9)) { ++j; return; }
        perturb >>= 5;
        base = (base * 5 + 1 + perturb) & mask;
        j = 0;
    }
};

__global__ void __launch_bounds__(256)
pyset_insert_kernel(const long long *__restrict__ hash, const u32 *__restrict__ ord, u32 n_re, u32 fresh_end, u64 mask,
                    u64 *__restrict__ tab) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= fresh_end) return;
    const u32 idx0 = t < n_re ? ord[t] : t;          // (the keys before n_re are exactly those already in the set)
    u64 mine = ((u64)t << 32) | idx0;
    PysetProbe pr;
    pr.start((u64)hash[idx0], mask);
    for (;;) {
        const u64 sl = pr.slot();
        const u64 old = atomicMin((unsigned long long *)&tab[sl], (unsigned long long)mine);
        if (old == PYSET_EMPTY) return;
        if (old > mine) {                            // displaced a later key: carry it on from this slot
            mine = old;
            pr.start((u64)hash[(u32)old], mask);
            while (pr.slot() != sl) pr.next();
        }
        pr.next();
    }
}

__global__ void __launch_bounds__(256)
pyset_flag_kernel(const u64 *__restrict__ tab, u64 size, u32 *__restrict__ flag) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= size) flag[i] = (i < size && tab[i] != PYSET_EMPTY) ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
pyset_list_kernel(const u64 *__restrict__ tab, u64 size, const u32 *__restrict__ at, u32 add, u32 *__restrict__ ord) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < size && tab[i] != PYSET_EMPTY) ord[at[i]] = (u32)tab[i] + add;
}

// d_hash[0 .. n): hashes in insertion order (distinct keys) -> d_order[0 .. n): `add` + the indices in the set's iteration order
static int pyset_order_device(catchhip_ctx *ctx, const long long *d_hash, u32 n, u32 add, u32 *d_order) {
    hipStream_t s = ctx->stream;
    // the last table: the generation the n-th insertion leaves behind
    std::vector<std::pair<u64, u32>> gens;           // (slots, keys in the set when this generation is complete)
    {
        u64 size = 8;
        u32 done = 0;
        for (;;) {
            const u64 mask = size - 1, first_rebuild = (mask * 3 + 4) / 5;
            done = (u32)std::min<u64>(n, first_rebuild);
            gens.push_back({size, done});
            if ((u64)done * 5 < mask * 3) break;
            const u64 want = done > 50000 ? (u64)done * 2 : (u64)done * 4;
            size = 8;
            while (size <= want) size <<= 1;
            if (done == n) { gens.push_back({size, done}); break; }
        }
    }
    const u64 last = gens.back().first;
    DevBuf<u64> tab;
    DevBuf<u32> flag, at, tmp, ord;
    TRY(tab.alloc((size_t)last));
    TRY(flag.alloc((size_t)last + 1));
    TRY(ord.alloc((size_t)n + 1));
    u32 n_re = 0;
    for (size_t g = 0; g < gens.size(); ++g) {
        const u64 size = gens[g].first;
        const u32 upto = gens[g].second;
        const bool final_gen = g + 1 == gens.size();
        HIP_TRY(hipMemsetAsync(tab.p, 0xff, sizeof(u64) * size, s));
        hipLaunchKernelGGL(pyset_insert_kernel, dim3((unsigned)div_up((i64)upto, 256)), dim3(256), 0, s,
                           d_hash, (const u32 *)ord.p, n_re, upto, size - 1, tab.p);
        hipLaunchKernelGGL(pyset_flag_kernel, dim3((unsigned)div_up((i64)size + 1, 256)), dim3(256), 0, s,
                           (const u64 *)tab.p, size, flag.p);
        TRY(cand_scan(ctx, flag, at, (i64)size + 1, tmp));
        hipLaunchKernelGGL(pyset_list_kernel, dim3((unsigned)div_up((i64)size, 256)), dim3(256), 0, s,
                           (const u64 *)tab.p, size, (const u32 *)at.p, final_gen ? add : 0u, final_gen ? d_order : ord.p);
        n_re = upto;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// (tests) hashes from the host, order back to the host
extern "C" int catchhip_pyset_order_device(catchhip_ctx *ctx, const i64 *hashes, i64 n, i64 *order) {
    ARG_CHECK(ctx && n >= 0 && n < ((i64)1 << 31) && (n == 0 || (hashes && order)));
    PoolScope pool_scope(ctx);
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<long long> d_hash;
    DevBuf<u32> d_order;
    TRY(d_hash.alloc((size_t)n));
    TRY(d_order.alloc((size_t)n + 1));
    HIP_TRY(hipMemcpyAsync(d_hash.p, hashes, sizeof(i64) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    TRY(pyset_order_device(ctx, d_hash.p, (u32)n, 0, d_order.p));
    std::vector<u32> h((size_t)n);
    HIP_TRY(hipMemcpyAsync(h.data(), d_order.p, sizeof(u32) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (i64 i = 0; i < n; ++i) order[i] = h[(size_t)i];
    return 0;
}

// The kept candidates as the reference hands them on: `list(to_include)`, a set of probes
// (near_duplicate_filter.py:76-103) -- per group (one _filter call each) the iteration order of a set
// the kept probes were added to in inclusion order (chip_pyset_order).  kept / kgrp: the kept candidates
// in inclusion order (groups non-decreasing).
static int cand_set_order(catchhip_ctx *ctx, const catchhip_candidates *C, DevBuf<u32> &kept, DevBuf<u32> &kgrp, u32 nk) {
    if (nk < 2) return 0;
    hipStream_t s = ctx->stream;
    DevBuf<long long> d_hash;
    TRY(d_hash.alloc(nk));
    hipLaunchKernelGGL(cand_pyhash_kernel, dim3((unsigned)div_up((i64)nk, 256)), dim3(256), 0, s,
                       (const u8 *)C->T->bytes.p, (const u32 *)kept.p, nk, (u32)C->L, d_hash.p);
    // groups: runs of kgrp.  Large ones are ordered on the device, the others by the host's emulation
    std::vector<u32> h_grp, bounds{0};
    if (C->grouped) {
        h_grp.resize(nk);
        HIP_TRY(hipMemcpyAsync(h_grp.data(), kgrp.p, sizeof(u32) * nk, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (u32 i = 1; i < nk; ++i)
            if (h_grp[i] != h_grp[i - 1]) bounds.push_back(i);
    }
    bounds.push_back(nk);
    static const u32 device_from = chip_test_env("CATCHHIP_PYSET_DEVICE_FROM") ? (u32)atoll(chip_test_env("CATCHHIP_PYSET_DEVICE_FROM")) : 8192u;
    bool any_host = false;
    for (size_t g = 0; g + 1 < bounds.size(); ++g) any_host |= bounds[g + 1] - bounds[g] < device_from;
    std::vector<i64> h_hash, order;
    std::vector<u32> o32;
    if (any_host) {
        h_hash.resize(nk);
        order.resize(nk);
        o32.resize(nk);
        HIP_TRY(hipMemcpyAsync(h_hash.data(), d_hash.p, sizeof(i64) * nk, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    DevBuf<u32> d_order, out;
    TRY(d_order.alloc(nk));
    TRY(out.alloc((size_t)nk + 1));
    for (size_t g = 0; g + 1 < bounds.size(); ++g) {
        const u32 a = bounds[g], b = bounds[g + 1];
        if (b - a >= device_from) {
            TRY(pyset_order_device(ctx, d_hash.p + a, b - a, a, d_order.p + a));
            continue;
        }
        chip_pyset_order(h_hash.data() + a, (i64)(b - a), order.data() + a);
        for (u32 i = a; i < b; ++i) o32[i] = (u32)(order[i] + a);
    }
    if (any_host) {       // the host-ordered runs, each to its place (usually all of them or none)
        for (size_t g = 0; g + 1 < bounds.size();) {
            if (bounds[g + 1] - bounds[g] >= device_from) { ++g; continue; }
            size_t e = g;
            while (e + 1 < bounds.size() && bounds[e + 1] - bounds[e] < device_from) ++e;
            HIP_TRY(hipMemcpyAsync(d_order.p + bounds[g], o32.data() + bounds[g], sizeof(u32) * (bounds[e] - bounds[g]),
                                   hipMemcpyHostToDevice, s));
            g = e;
        }
    }
    hipLaunchKernelGGL(cand_permute_kernel, dim3((unsigned)div_up((i64)nk, 256)), dim3(256), 0, s,
                       (const u32 *)kept.p, (const u32 *)d_order.p, nk, out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));     // (o32 is read by the copies until here)
    kept.swap(out);                       // groups: a permutation inside every group leaves kgrp as it is
    return 0;
}

// flag[0..n) (device, 0 / 1; n + 1 words allocated) -> the candidate list becomes the kept ones, in the reference's
// order (see cand_set_order)
static int cand_apply_keep(catchhip_ctx *ctx, catchhip_candidates *C, DevBuf<u32> &opos, DevBuf<u32> &ogrp,
                           DevBuf<u32> &flag, i64 *nkept) {
    hipStream_t s = ctx->stream;
    const u32 n = (u32)C->nuniq;
    DevBuf<u32> at, tmp, out;
    HIP_TRY(hipMemsetAsync(flag.p + n, 0, sizeof(u32), s));
    TRY(cand_scan(ctx, flag, at, (i64)n + 1, tmp));
    u32 nk = 0;
    TRY(cand_read_u32(ctx, at.p + n, &nk));
    TRY(out.alloc((size_t)nk + 1));
    hipLaunchKernelGGL(cand_compact_kernel, dim3((unsigned)div_up((i64)n, 256)), dim3(256), 0, s,
                       (const u32 *)opos.p, (const u32 *)flag.p, (const u32 *)at.p, n, out.p);
    DevBuf<u32> gout;
    if (C->grouped) {
        TRY(gout.alloc((size_t)nk + 1));
        hipLaunchKernelGGL(cand_compact_kernel, dim3((unsigned)div_up((i64)n, 256)), dim3(256), 0, s,
                           (const u32 *)ogrp.p, (const u32 *)flag.p, (const u32 *)at.p, n, gout.p);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    if (!chip_test_env("CATCHHIP_NDF_INCLUSION_ORDER")) TRY(cand_set_order(ctx, C, out, gout, nk));
    C->upos.swap(out);
    if (C->grouped) C->ugrp.swap(gout);
    C->nuniq = nk;
    C->filtered = true;
    if (nkept) *nkept = nk;
    return 0;
}

static int cand_ndf_hamming(catchhip_ctx *ctx, catchhip_candidates *C, const i32 *positions, i64 ngroups,
                            i32 ntables, i32 k, i32 dist_thres, i64 *nkept) {
    ARG_CHECK(ctx && C && C->ctx == ctx && positions && ntables >= 1 && k >= 1);
    PoolScope pool_scope(ctx);
    if (C->filtered) { chip_set_error("candidates: a near-duplicate filter was already applied"); return CATCHHIP_EINVAL; }
    HIP_TRY(hipSetDevice(ctx->device));
    if (C->nuniq == 0) { C->filtered = true; if (nkept) *nkept = 0; return 0; }
    DevBuf<u32> opos, ogrp;
    DevBuf<u8> rows;
    TRY(cand_priority_rows(ctx, C, opos, rows, ogrp));
    DevBuf<u32> flag;      // the verdicts stay on the device (round 5)
    TRY(flag.alloc((size_t)C->nuniq + 1));
    TRY(chip_ndf_hamming_device(ctx, rows.p, C->nuniq, C->L, positions, ntables, k, dist_thres, nullptr,
                                C->grouped ? (const u32 *)ogrp.p : (const u32 *)nullptr, C->grouped ? ngroups : 1, flag.p));
    return cand_apply_keep(ctx, C, opos, ogrp, flag, nkept);
}

extern "C" int catchhip_candidates_ndf_hamming(catchhip_ctx *ctx, catchhip_candidates *C, const i32 *positions,
                                               i32 ntables, i32 k, i32 dist_thres, i64 *nkept) {
    ARG_CHECK(C);
    if (C->grouped) {
        chip_set_error("candidates_ndf_hamming: grouped candidates need one set of sampled positions per group (..._many)");
        return CATCHHIP_EINVAL;
    }
    return cand_ndf_hamming(ctx, C, positions, 1, ntables, k, dist_thres, nkept);
}

extern "C" int catchhip_candidates_ndf_hamming_many(catchhip_ctx *ctx, catchhip_candidates *C, const i32 *positions,
                                                    i64 ngroups, i32 ntables, i32 k, i32 dist_thres, i64 *nkept) {
    ARG_CHECK(C);
    if (!C->grouped || ngroups < C->ngroups) {
        chip_set_error("candidates_ndf_hamming_many: the targets carry no groups, or more groups than position sets");
        return CATCHHIP_EINVAL;
    }
    return cand_ndf_hamming(ctx, C, positions, ngroups, ntables, k, dist_thres, nkept);
}

static int cand_ndf_minhash(catchhip_ctx *ctx, catchhip_candidates *C, i32 kmer_size, const i64 *ab, i64 ngroups,
                            i32 ntables, i32 k, double dist_thres, i64 *nkept) {
    ARG_CHECK(ctx && C && C->ctx == ctx && ab && ntables >= 1 && k >= 1);
    PoolScope pool_scope(ctx);
    if (C->filtered) { chip_set_error("candidates: a near-duplicate filter was already applied"); return CATCHHIP_EINVAL; }
    HIP_TRY(hipSetDevice(ctx->device));
    if (C->nuniq == 0) { C->filtered = true; if (nkept) *nkept = 0; return 0; }
    DevBuf<u32> opos, ogrp;
    DevBuf<u8> rows;
    TRY(cand_priority_rows(ctx, C, opos, rows, ogrp));
    // equal-length rows, their groups (runs of the priority order) and the verdicts all stay on the device (round 5)
    if (C->grouped) ARG_CHECK(ngroups >= C->ngroups);
    DevBuf<u32> flag;
    TRY(flag.alloc((size_t)C->nuniq + 1));
    TRY(chip_ndf_minhash_rows(ctx, rows.p, C->nuniq, C->L, C->grouped ? (const u32 *)ogrp.p : (const u32 *)nullptr,
                              C->grouped ? ngroups : 1, kmer_size, ab, ntables, k, dist_thres, flag.p));
    return cand_apply_keep(ctx, C, opos, ogrp, flag, nkept);
}

extern "C" int catchhip_candidates_ndf_minhash(catchhip_ctx *ctx, catchhip_candidates *C, i32 kmer_size,
                                               const i64 *ab, i32 ntables, i32 k, double dist_thres, i64 *nkept) {
    ARG_CHECK(C);
    if (C->grouped) {
        chip_set_error("candidates_ndf_minhash: grouped candidates need one set of hash functions per group (..._many)");
        return CATCHHIP_EINVAL;
    }
    return cand_ndf_minhash(ctx, C, kmer_size, ab, 1, ntables, k, dist_thres, nkept);
}

extern "C" int catchhip_candidates_ndf_minhash_many(catchhip_ctx *ctx, catchhip_candidates *C, i32 kmer_size,
                                                    const i64 *ab, i64 ngroups, i32 ntables, i32 k,
                                                    double dist_thres, i64 *nkept) {
    ARG_CHECK(C);
    if (!C->grouped || ngroups < C->ngroups) {
        chip_set_error("candidates_ndf_minhash_many: the targets carry no groups, or more groups than hash function sets");
        return CATCHHIP_EINVAL;
    }
    return cand_ndf_minhash(ctx, C, kmer_size, ab, ngroups, ntables, k, dist_thres, nkept);
}

extern "C" int catchhip_candidates_groups(catchhip_ctx *ctx, const catchhip_candidates *C, i32 *group_of_candidate) {
    ARG_CHECK(ctx && C && C->ctx == ctx);
    if (C->nuniq == 0) return 0;
    ARG_CHECK(group_of_candidate);
    if (!C->grouped) { memset(group_of_candidate, 0, sizeof(i32) * (size_t)C->nuniq); return 0; }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(group_of_candidate, C->ugrp.p, sizeof(i32) * (size_t)C->nuniq, hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// Random anchors from their DRAWS (round 5): draws[probe][m] = the m positions np.random drew for the probe
// (catch/probe.py:356-405: 20 per probe, repeats allowed).  The table is the sorted distinct positions of every probe --
// a 256-bit set per probe on the device instead of a NumPy row sort, boolean masks and 8-byte entries on the host
// (3.5 x the time of the draws themselves, and 570 MB of pageable uploads per 2 M probes).
__global__ void __launch_bounds__(256)
cand_draws_count_kernel(const u8 *__restrict__ draws, u32 n, u32 m, u32 *__restrict__ cnt) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    u32 c = 0;
    if (i < n) {
        u64 b[4] = {0, 0, 0, 0};
        for (u32 q = 0; q < m; ++q) { const u32 v = draws[(size_t)i * m + q]; b[v >> 6] |= 1ull << (v & 63u); }
        c = (u32)(__popcll(b[0]) + __popcll(b[1]) + __popcll(b[2]) + __popcll(b[3]));
    }
    cnt[i] = c;       // (entry n: the sentinel of the prefix sum)
}
__global__ void __launch_bounds__(256)
cand_draws_fill_kernel(const u8 *__restrict__ draws, u32 n, u32 m, const u32 *__restrict__ ptr, i32 *__restrict__ ent_probe,
                       i32 *__restrict__ ent_pos, u32 *__restrict__ sent_probe, u32 *__restrict__ sent_pos) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 b[4] = {0, 0, 0, 0};
    for (u32 q = 0; q < m; ++q) { const u32 v = draws[(size_t)i * m + q]; b[v >> 6] |= 1ull << (v & 63u); }
    u32 o = ptr[i];
    for (u32 w = 0; w < 4; ++w) {
        u64 x = b[w];
        while (x) {
            const u32 v = w * 64u + (u32)(__ffsll((long long)x) - 1);
            x &= x - 1ull;
            ent_probe[o] = (i32)i; ent_pos[o] = (i32)v;
            sent_probe[o] = i; sent_pos[o] = v;
            ++o;
        }
    }
}

static int probes_from_candidates_impl(catchhip_ctx *ctx, const catchhip_candidates *C, const i32 *ent_probe,
                                       const i32 *ent_pos, i64 nent, i32 k, catchhip_probes **out, const u8 *draws, i32 m) {
    ARG_CHECK(ctx && C && C->ctx == ctx && out && k > 0 && k <= C->L);
    ARG_CHECK((ent_probe == nullptr) == (ent_pos == nullptr));
    PoolScope pool_scope(ctx);
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    const i64 n = C->nuniq, L = C->L;
    ARG_CHECK(n * L < ((i64)1 << 31));
    const bool pigeon = ent_probe == nullptr && draws == nullptr;
    DevBuf<u8> d_draws;
    DevBuf<u32> d_cnt, d_ptr, d_tmp;
    if (draws) {
        // the anchors' count first (the probes object is sized by it): distinct draws per probe, prefix sum
        ARG_CHECK(m >= 1 && L - k + 1 <= 256 && n * (i64)m < ((i64)1 << 32));
        for (i64 e = 0; e < n * m; ++e) ARG_CHECK((i64)draws[e] + k <= L);
        TRY(d_draws.alloc((size_t)(n * m) + 1));
        TRY(d_cnt.alloc((size_t)n + 1));
        TRY(chip_pinned_reserve(ctx, (size_t)(n * m) + 1));
        memcpy(ctx->h_big, draws, (size_t)(n * m));
        HIP_TRY(hipMemcpyAsync(d_draws.p, ctx->h_big, (size_t)(n * m), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(cand_draws_count_kernel, dim3((unsigned)div_up(n + 1, 256)), dim3(256), 0, ctx->stream,
                           (const u8 *)d_draws.p, (u32)n, (u32)m, d_cnt.p);
        TRY(cand_scan(ctx, d_cnt, d_ptr, n + 1, d_tmp));
        u32 total_ent = 0;
        TRY(cand_read_u32(ctx, d_ptr.p + n, &total_ent));
        nent = total_ent;
    } else if (pigeon) {
        ARG_CHECK(L % k == 0);
        nent = n * (L / k);
    } else {
        ARG_CHECK(nent >= 0);
        for (i64 e = 0; e < nent; ++e) {
            ARG_CHECK(ent_probe[e] >= 0 && ent_probe[e] < n && ent_pos[e] >= 0 && ent_pos[e] + k <= L);
            if (e && (ent_probe[e] < ent_probe[e - 1] ||
                      (ent_probe[e] == ent_probe[e - 1] && ent_pos[e] <= ent_pos[e - 1]))) {
                chip_set_error("probes_from_candidates: anchors must be sorted by (probe, position), no duplicates");
                return CATCHHIP_EINVAL;
            }
        }
    }
    catchhip_probes *p = new catchhip_probes();
    p->ctx = ctx;
    p->nprobes = n;
    p->total = n * L;
    p->nent = nent;
    p->k = k;
    p->L = n ? (i32)L : 0;
    p->max_set_id = n ? n - 1 : 0;
    p->nbuckets = n;
    p->dna5 = C->T->dna5;       // windows of the targets: a subset of their alphabet
    p->has_n = C->T->has_n;
    p->sorted_unique = true;
    hipStream_t s = ctx->stream;
    int rc = 0;
    do {
        if ((rc = p->bytes.alloc((size_t)p->total + 256)) || (rc = p->probe_off.alloc((size_t)n + 1)) ||
            (rc = p->set_id.alloc((size_t)n + 1)) || (rc = p->bucket_of.alloc((size_t)n + 1)) ||
            (rc = p->bucket_set.alloc((size_t)n + 1)) || (rc = p->ent_probe.alloc((size_t)nent + 1)) ||
            (rc = p->ent_pos.alloc((size_t)nent + 1)) || (rc = p->sent_probe.alloc((size_t)nent + 1)) ||
            (rc = p->sent_pos.alloc((size_t)nent + 1)) || (rc = p->ent_ptr.alloc((size_t)n + 1)))
            break;
        if (hipMemsetAsync(p->bytes.p, 0, (size_t)p->total + 256, s) != hipSuccess) { rc = CATCHHIP_EHIP; break; }
        hipLaunchKernelGGL(cand_gather_kernel, dim3((unsigned)div_up(std::max<i64>(p->total, n + 1), 256)), dim3(256), 0,
                           s, (const u8 *)C->T->bytes.p, (const u32 *)C->upos.p, (u32)n, (u32)L, p->bytes.p,
                           p->probe_off.p, p->set_id.p, p->bucket_of.p, p->bucket_set.p);
        p->bucket_identity = true;
        if (draws) {
            p->pigeonhole = false;
            if (n) {
                hipLaunchKernelGGL(cand_draws_fill_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, s, (const u8 *)d_draws.p,
                                   (u32)n, (u32)m, (const u32 *)d_ptr.p, p->ent_probe.p, p->ent_pos.p, p->sent_probe.p, p->sent_pos.p);
            }
            if (hipMemcpyAsync(p->ent_ptr.p, d_ptr.p, sizeof(u32) * ((size_t)n + 1), hipMemcpyDeviceToDevice, s) != hipSuccess) {
                rc = CATCHHIP_EHIP;
                break;
            }
        } else if (pigeon) {
            const u32 nanch = (u32)(L / k);
            p->pigeonhole = n > 0;
            hipLaunchKernelGGL(cand_pigeon_kernel, dim3((unsigned)div_up(std::max<i64>(nent, n + 1), 256)), dim3(256), 0,
                               s, (u32)n, nanch, (u32)k, p->ent_probe.p, p->ent_pos.p, p->sent_probe.p, p->sent_pos.p,
                               p->ent_ptr.p);
        } else {
            // given sorted by (probe, position): the sorted table is the table
            std::vector<u32> ptr((size_t)n + 1, 0);
            bool pg = n > 0 && L % k == 0 && nent == n * (L / k);
            for (i64 e = 0; e < nent; ++e) {
                ptr[(size_t)ent_probe[e] + 1]++;
                if (pg && ent_pos[e] != (i32)((e % (L / k)) * k)) pg = false;
            }
            for (i64 i = 0; i < n; ++i) {
                if (pg && ptr[(size_t)i + 1] != (u32)(L / k)) pg = false;
                ptr[(size_t)i + 1] += ptr[(size_t)i];
            }
            p->pigeonhole = pg;
            if (nent && (hipMemcpyAsync(p->ent_probe.p, ent_probe, sizeof(i32) * nent, hipMemcpyHostToDevice, s) != hipSuccess ||
                         hipMemcpyAsync(p->ent_pos.p, ent_pos, sizeof(i32) * nent, hipMemcpyHostToDevice, s) != hipSuccess ||
                         hipMemcpyAsync(p->sent_probe.p, ent_probe, sizeof(i32) * nent, hipMemcpyHostToDevice, s) != hipSuccess ||
                         hipMemcpyAsync(p->sent_pos.p, ent_pos, sizeof(i32) * nent, hipMemcpyHostToDevice, s) != hipSuccess)) {
                rc = CATCHHIP_EHIP;
                break;
            }
            if (hipMemcpyAsync(p->ent_ptr.p, ptr.data(), sizeof(u32) * ((size_t)n + 1), hipMemcpyHostToDevice, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess) {
                rc = CATCHHIP_EHIP;
                break;
            }
        }
        if (C->grouped) {   // the scans pair a probe only with its own group's genomes
            if ((rc = p->group.alloc((size_t)n + 1))) break;
            if (n && hipMemcpyAsync(p->group.p, C->ugrp.p, sizeof(u32) * (size_t)n, hipMemcpyDeviceToDevice, s) != hipSuccess) {
                rc = CATCHHIP_EHIP;
                break;
            }
            p->has_groups = true;
        }
        if ((rc = chip_probes_pack_planes(p))) break;
        if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) {
            chip_set_error("probes_from_candidates failed: %s", hipGetErrorString(hipGetLastError()));
            rc = CATCHHIP_EHIP;
            break;
        }
    } while (0);
    if (rc) { delete p; return rc; }
    *out = p;
    return 0;
}

extern "C" int catchhip_probes_from_candidates(catchhip_ctx *ctx, const catchhip_candidates *C, const i32 *ent_probe,
                                               const i32 *ent_pos, i64 nent, i32 k, catchhip_probes **out) {
    return probes_from_candidates_impl(ctx, C, ent_probe, ent_pos, nent, k, out, nullptr, 0);
}

extern "C" int catchhip_probes_from_candidates_draws(catchhip_ctx *ctx, const catchhip_candidates *C, const u8 *draws,
                                                     i32 draws_per_probe, i32 k, catchhip_probes **out) {
    ARG_CHECK(draws || (C && C->nuniq == 0));
    if (!draws) return probes_from_candidates_impl(ctx, C, nullptr, nullptr, 0, k, out, (const u8 *)"", 1);
    return probes_from_candidates_impl(ctx, C, nullptr, nullptr, 0, k, out, draws, draws_per_probe);
}
