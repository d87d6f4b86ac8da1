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

#include "internal.h"

__global__ void __launch_bounds__(256)
ndf_key_kernel(const u8 *__restrict__ bytes, u32 n, int L, const i32 *__restrict__ pos, int k,
               u64 *__restrict__ keys, u32 *__restrict__ vals) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u8 *p = bytes + (size_t)i * L;
    u64 h = 0xcbf29ce484222325ull;
    for (int j = 0; j < k; ++j) h = (h ^ (u64)p[pos[j]]) * 0x100000001b3ull;
    keys[i] = h;
    vals[i] = i;
}

__device__ __forceinline__ bool ndf_near(const u8 *__restrict__ a, const u8 *__restrict__ b, int L, int d,
                                         const i32 *__restrict__ pos, int k) {
    int mm = 0;
    for (int j = 0; j < L; ++j) {
        mm += (a[j] != b[j]);
        if (mm > d) return false;
    }
    for (int j = 0; j < k; ++j)
        if (a[pos[j]] != b[pos[j]]) return false;  // different bucket (hash collision)
    return true;
}

__global__ void __launch_bounds__(256)
ndf_edge_kernel(const u8 *__restrict__ bytes, u32 n, int L, int d, const i32 *__restrict__ pos, int k,
                const u64 *__restrict__ keys, const u32 *__restrict__ vals, u32 *__restrict__ e_i,
                u32 *__restrict__ e_j, u32 *__restrict__ count, u32 cap) {
    u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const u64 key = keys[x];
    const u32 i = vals[x];
    const u8 *a = bytes + (size_t)i * L;
    for (u32 y = x; y-- > 0;) {
        if (keys[y] != key) break;
        const u32 j = vals[y];  // j < i: stable sort keeps indices ascending in a run
        if (ndf_near(a, bytes + (size_t)j * L, L, d, pos, k)) {
            u32 slot = atomicAdd(count, 1u);
            if (slot < cap) { e_i[slot] = i; e_j[slot] = j; }
        }
    }
}

// status: 0 undecided, 1 kept, 2 dropped.  flags: bit0 = has kept higher
// neighbour, bit1 = has undecided higher neighbour
__global__ void __launch_bounds__(256)
ndf_edge_round_kernel(const u32 *__restrict__ e_i, const u32 *__restrict__ e_j, u32 ne,
                      const u32 *__restrict__ status, u32 *__restrict__ flags) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ne) return;
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

extern "C" int catchhip_ndf_hamming(catchhip_ctx *ctx, const u8 *bytes, i64 n, i32 L, const i32 *positions,
                                    i32 ntables, i32 k, i32 dist_thres, u8 *keep) {
    ARG_CHECK(ctx && n >= 0 && L > 0 && ntables >= 1 && k >= 1 && positions);
    PoolScope pool_scope(ctx);
    if (n == 0) return 0;
    ARG_CHECK(bytes && keep);
    ARG_CHECK(n < ((i64)1 << 31) && n * (i64)L < ((i64)1 << 40));
    for (i64 t = 0; t < (i64)ntables * k; ++t) ARG_CHECK(positions[t] >= 0 && positions[t] < L);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const u32 nn = (u32)n;
    DevBuf<u8> d_bytes;
    DevBuf<i32> d_pos;
    DevBuf<u64> keys, keys_alt;
    DevBuf<u32> vals, vals_alt, e_i, e_j, count, status, flags;
    TRY(d_bytes.alloc((size_t)n * L));
    TRY(d_pos.alloc((size_t)ntables * k));
    TRY(keys.alloc(nn));
    TRY(vals.alloc(nn));
    TRY(count.alloc(2));
    TRY(status.alloc(nn));
    TRY(flags.alloc(nn));
    HIP_TRY(hipMemcpyAsync(d_bytes.p, bytes, (size_t)n * L, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_pos.p, positions, sizeof(i32) * ntables * k, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(count.p, 0, 2 * sizeof(u32), s));
    HIP_TRY(hipMemsetAsync(status.p, 0, sizeof(u32) * nn, s));
    HIP_TRY(hipMemsetAsync(flags.p, 0, sizeof(u32) * nn, s));

    PhaseTimer tm(ctx, PHASE_NDF);
    const unsigned nb = (unsigned)div_up(nn, 256);
    u32 cap = (u32)std::max<i64>((i64)1 << 20, std::min<i64>(n * 16, (i64)1 << 28));
    u32 ne = 0;
    for (int attempt = 0;; ++attempt) {
        TRY(e_i.reserve(cap));
        TRY(e_j.reserve(cap));
        HIP_TRY(hipMemsetAsync(count.p, 0, sizeof(u32), s));
        for (int t = 0; t < ntables; ++t) {
            hipLaunchKernelGGL(ndf_key_kernel, dim3(nb), dim3(256), 0, s, d_bytes.p, nn, (int)L,
                               d_pos.p + (size_t)t * k, (int)k, keys.p, vals.p);
            TRY(chip_radix_sort_pairs(ctx, keys, keys_alt, vals, vals_alt, nn, 64));
            hipLaunchKernelGGL(ndf_edge_kernel, dim3(nb), dim3(256), 0, s, d_bytes.p, nn, (int)L,
                               (int)dist_thres, d_pos.p + (size_t)t * k, (int)k, keys.p, vals.p, e_i.p,
                               e_j.p, count.p, cap);
            tm.launch(2 + 24);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ctx->h_pin, count.p, sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        ne = *(volatile u32 *)ctx->h_pin;
        if (ne <= cap) break;
        if (attempt >= 2) { chip_set_error("ndf: edge buffer overflow"); return CATCHHIP_ENOMEM; }
        cap = ne;
    }
    // greedy resolution rounds
    for (u32 round = 0; round <= nn + 1; ++round) {
        HIP_TRY(hipMemsetAsync(count.p + 1, 0, sizeof(u32), s));
        if (ne)
            hipLaunchKernelGGL(ndf_edge_round_kernel, dim3((unsigned)div_up(ne, 256)), dim3(256), 0, s, e_i.p,
                               e_j.p, ne, status.p, flags.p);
        hipLaunchKernelGGL(ndf_node_round_kernel, dim3(nb), dim3(256), 0, s, status.p, flags.p, nn, count.p + 1);
        tm.launch(2);
        HIP_TRY(hipMemcpyAsync(ctx->h_pin, count.p + 1, sizeof(u32), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (*(volatile u32 *)ctx->h_pin == 0) break;
    }
    tm.stop();
    std::vector<u32> h_status(nn);
    HIP_TRY(hipMemcpyAsync(h_status.data(), status.p, sizeof(u32) * nn, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    tm.finish();
    for (u32 i = 0; i < nn; ++i) {
        if (h_status[i] == 0) { chip_set_error("ndf: unresolved probe"); return CATCHHIP_EINVAL; }
        keep[i] = h_status[i] == 1 ? 1 : 0;
    }
    return 0;
}
