// Device-wide primitives used by the cover-row builder and the near-duplicate
// filter: exclusive prefix sum and a stable LSD radix sort of (u64 key, u32
// value) pairs.  Hand-written for gfx950: 64-lane wavefronts, wave-level
// multi-split through __ballot, LDS for the per-workgroup digit tables.
#include "internal.h"

// ------------------------------------------------------------------------
// exclusive scan (u32): tile reduce -> recursive scan of tile sums -> tile scan
// ------------------------------------------------------------------------
#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        u32 t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix,
// *block_total = sum over the block.  lds: 4+1 words.
__device__ __forceinline__ u32 block_excl_scan_u32(u32 v, u32 *lds, u32 *block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 inc = wave_incl_scan_u32(v, lane);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    u32 woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / WAVE; ++w) {
        u32 t = lds[w];
        if (w < wave) woff += t;
        tot += t;
    }
    __syncthreads();
    *block_total = tot;
    return woff + inc - v;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_tile_reduce(const u32 *__restrict__ in, u32 *__restrict__ tile_sums, i64 n) {
    __shared__ u32 lds[8];
    i64 base = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_ITEMS;
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)
        if (base + j < n) s += in[base + j];
    u32 tot;
    (void)block_excl_scan_u32(s, lds, &tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_tile_apply(const u32 *__restrict__ in, u32 *__restrict__ out,
                const u32 *__restrict__ tile_offs, i64 n) {
    __shared__ u32 lds[8];
    i64 base = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS];
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        v[j] = (base + j < n) ? in[base + j] : 0u;
        s += v[j];
    }
    u32 tot;
    u32 ex = block_excl_scan_u32(s, lds, &tot);
    u32 run = ex + (tile_offs ? tile_offs[blockIdx.x] : 0u);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < n) out[base + j] = run;
        run += v[j];
    }
}

static int scan_rec(catchhip_ctx *ctx, const u32 *in, u32 *out, i64 n, u32 *tmp, i64 tmp_n) {
    if (n <= 0) return 0;
    i64 nt = div_up(n, SCAN_TILE);
    if (nt == 1) {
        hipLaunchKernelGGL(scan_tile_apply, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, in, out,
                           (const u32 *)nullptr, n);
        return 0;
    }
    if (tmp_n < nt) {
        chip_set_error("scan: scratch too small");
        return CATCHHIP_EINVAL;
    }
    hipLaunchKernelGGL(scan_tile_reduce, dim3((unsigned)nt), dim3(SCAN_THREADS), 0, ctx->stream, in,
                       tmp, n);
    TRY(scan_rec(ctx, tmp, tmp, nt, tmp + nt, tmp_n - nt));
    hipLaunchKernelGGL(scan_tile_apply, dim3((unsigned)nt), dim3(SCAN_THREADS), 0, ctx->stream, in,
                       out, (const u32 *)tmp, n);
    return 0;
}

int chip_exclusive_scan_u32(catchhip_ctx *ctx, const u32 *in, u32 *out, i64 n, DevBuf<u32> &tmp) {
    if (n <= 0) return 0;
    i64 need = 0;
    for (i64 m = div_up(n, SCAN_TILE); m > 1; m = div_up(m, SCAN_TILE)) need += m;
    need += 4;
    TRY(tmp.reserve((size_t)need));
    TRY(scan_rec(ctx, in, out, n, tmp.p, (i64)tmp.n));
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------
// LSD radix sort, 8-bit digits, stable.
// ------------------------------------------------------------------------
#define RS_THREADS 256
#define RS_ROUNDS 16
#define RS_TILE (RS_THREADS * RS_ROUNDS)

__global__ void __launch_bounds__(RS_THREADS)
radix_hist(const u64 *__restrict__ keys, i64 n, int shift, u32 *__restrict__ hist, u32 nblocks) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    i64 base = (i64)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int r = 0; r < RS_ROUNDS; ++r) {
        i64 idx = base + (i64)r * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(u32)(keys[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(RS_THREADS)
radix_scatter(const u64 *__restrict__ keys_in, const u32 *__restrict__ vals_in,
              u64 *__restrict__ keys_out, u32 *__restrict__ vals_out, i64 n, int shift,
              const u32 *__restrict__ hist_scanned, u32 nblocks) {
    __shared__ u32 wave_cnt[RS_THREADS / WAVE][256];
    __shared__ u32 base[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    base[tid] = hist_scanned[(size_t)tid * nblocks + blockIdx.x];
    const i64 tile0 = (i64)blockIdx.x * RS_TILE;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < RS_ROUNDS; ++r) {
        i64 idx = tile0 + (i64)r * RS_THREADS + tid;
        if (tile0 + (i64)r * RS_THREADS >= n) break;  // uniform
        bool valid = idx < n;
        u64 key = valid ? keys_in[idx] : 0ull;
        u32 val = valid ? vals_in[idx] : 0u;
        u32 digit = (u32)(key >> shift) & 255u;
#pragma unroll
        for (int w = 0; w < RS_THREADS / WAVE; ++w) wave_cnt[w][tid] = 0;
        __syncthreads();
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            bool bit = (digit >> b) & 1u;
            u64 bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        u32 rank_in_wave = (u32)__popcll(peers & lt_mask);
        if (valid && rank_in_wave == 0) wave_cnt[wave][digit] = (u32)__popcll(peers);
        __syncthreads();
        {
            u32 b = base[tid];
#pragma unroll
            for (int w = 0; w < RS_THREADS / WAVE; ++w) {
                u32 c = wave_cnt[w][tid];
                wave_cnt[w][tid] = b;
                b += c;
            }
            base[tid] = b;
        }
        __syncthreads();
        if (valid) {
            u32 pos = wave_cnt[wave][digit] + rank_in_wave;
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
    }
}

int chip_radix_sort_pairs(catchhip_ctx *ctx, DevBuf<u64> &keys, DevBuf<u64> &keys_alt,
                          DevBuf<u32> &vals, DevBuf<u32> &vals_alt, i64 n, int key_bits, int first_bit) {
    if (n <= 1) return 0;
    if (n >= ((i64)1 << 32)) {
        chip_set_error("radix sort: n too large");
        return CATCHHIP_EINVAL;
    }
    TRY(keys_alt.reserve((size_t)n));
    TRY(vals_alt.reserve((size_t)n));
    u32 nblocks = (u32)div_up(n, RS_TILE);
    DevBuf<u32> hist, tmp;
    TRY(hist.alloc((size_t)256 * nblocks));
    int passes = (key_bits + 7) / 8;
    if (passes < 1) passes = 1;
    for (int p = 0; p < passes; ++p) {
        int shift = first_bit + 8 * p;
        hipLaunchKernelGGL(radix_hist, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, keys.p, n,
                           shift, hist.p, nblocks);
        TRY(chip_exclusive_scan_u32(ctx, hist.p, hist.p, (i64)256 * nblocks, tmp));
        hipLaunchKernelGGL(radix_scatter, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, keys.p,
                           vals.p, keys_alt.p, vals_alt.p, n, shift, hist.p, nblocks);
        keys.swap(keys_alt);
        vals.swap(vals_alt);
    }
    HIP_TRY(hipGetLastError());
    // scratch (hist/tmp) returns to the pool; reuse is ordered by the stream
    return 0;
}
