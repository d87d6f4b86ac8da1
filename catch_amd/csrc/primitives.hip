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
    // (blockIdx.y = segment of a segmented sort: segments of n keys side by side, every segment sorted on its own)
    keys += (size_t)blockIdx.y * n;
    hist += (size_t)blockIdx.y * 256 * nblocks;
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

// One pass of the scatter (round 4: tile-local sort first).  Rounds 1-3 ranked 256 keys per workgroup step and
// wrote every key straight to its place: ~1 key per digit and step, i.e. a lone 8-byte and a lone 4-byte write per
// key (134 M records of S4i's hit list: 2.6 ms per pass, ~1.2 TB/s of useful bytes).  Now a wavefront ranks its own
// 1,024 consecutive keys of the tile with wave-private digit counters (no workgroup barrier inside the 16 rounds),
// the tile is put in digit order in LDS, and the keys of one digit leave as one contiguous run.  Stable: a digit's
// keys keep the order (wavefront, round, lane) = their order in the input.
#define RS_WAVES (RS_THREADS / WAVE)
#define RS_PER_WAVE (RS_TILE / RS_WAVES)
__global__ void __launch_bounds__(RS_THREADS)
radix_scatter(const u64 *__restrict__ keys_in, const u32 *__restrict__ vals_in,
              u64 *__restrict__ keys_out, u32 *__restrict__ vals_out, i64 n, int shift,
              const u32 *__restrict__ hist_scanned, u32 nblocks) {
    __shared__ u64 s_key[RS_TILE];
    __shared__ u32 s_val[RS_TILE];
    __shared__ u32 wave_cnt[RS_WAVES][256];     // per wavefront: keys of each digit (then: its first slot in the tile order)
    __shared__ u32 dig_off[257];                // first tile slot of every digit
    __shared__ u32 gbase[256];                  // first output slot of every digit for this tile
    __shared__ u32 s_scan[RS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i64 tile0 = (i64)blockIdx.x * RS_TILE;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    {   // segment blockIdx.y: its keys, and its part of the histogram (scanned over ALL segments: minus the keys before it)
        const size_t seg = blockIdx.y;
        keys_in += seg * (size_t)n; vals_in += seg * (size_t)n; keys_out += seg * (size_t)n; vals_out += seg * (size_t)n;
        hist_scanned += seg * 256 * nblocks;
    }
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) wave_cnt[w][tid] = 0;
    gbase[tid] = hist_scanned[(size_t)tid * nblocks + blockIdx.x] - (u32)((size_t)blockIdx.y * (size_t)n);
    __syncthreads();
    u64 key[RS_ROUNDS];
    u32 val[RS_ROUNDS], rnk[RS_ROUNDS];       // rnk: rank among this wavefront's keys of the same digit
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const i64 idx = tile0 + (i64)wave * RS_PER_WAVE + (i64)r * WAVE + lane;
        const bool valid = idx < n;
        key[r] = valid ? keys_in[idx] : ~0ull;
        val[r] = valid ? vals_in[idx] : 0u;
        const u32 digit = (u32)(key[r] >> shift) & 255u;
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (digit >> b) & 1u;
            const u64 bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const u32 before = wave_cnt[wave][digit];          // (wave-private: LDS operations of one wavefront stay in order)
        rnk[r] = valid ? before + (u32)__popcll(peers & lt_mask) : 0xffffffffu;
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt_mask) == 0ull) wave_cnt[wave][digit] = before + (u32)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // thread tid = digit tid: its count in the tile, the wavefronts' shares of it, the exclusive scan over the digits
        u32 tot = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) { const u32 c = wave_cnt[w][tid]; wave_cnt[w][tid] = tot; tot += c; }
        u32 inc = wave_incl_scan_u32(tot, lane);
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        u32 woff = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) if (w < wave) woff += s_scan[w];
        dig_off[tid] = woff + inc - tot;
        if (tid == 255) dig_off[256] = woff + inc;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        if (rnk[r] != 0xffffffffu) {
            const u32 digit = (u32)(key[r] >> shift) & 255u;
            const u32 q = dig_off[digit] + wave_cnt[wave][digit] + rnk[r];
            s_key[q] = key[r];
            s_val[q] = val[r];
        }
    }
    __syncthreads();
    const u32 ntile = dig_off[256];
    for (u32 q = tid; q < ntile; q += RS_THREADS) {
        const u64 k = s_key[q];
        const u32 digit = (u32)(k >> shift) & 255u;
        const u32 pos = gbase[digit] + (q - dig_off[digit]);
        keys_out[pos] = k;
        vals_out[pos] = s_val[q];
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

// The same for `nseg` segments of n keys each, side by side in keys / vals (round 5: the 25 tables of a MinHash filter
// call in one go -- 350 sorts of ~10 launches per S5 step were 11,000 small launches queued behind other streams' kernels).
int chip_radix_sort_pairs_segments(catchhip_ctx *ctx, DevBuf<u64> &keys, DevBuf<u64> &keys_alt, DevBuf<u32> &vals,
                                   DevBuf<u32> &vals_alt, i64 n, i64 nseg, int key_bits, int first_bit) {
    if (n <= 1 || nseg <= 0) return 0;
    if (n * nseg >= ((i64)1 << 32) || nseg > 65535) {
        chip_set_error("radix sort: too many keys in all segments");
        return CATCHHIP_EINVAL;
    }
    TRY(keys_alt.reserve((size_t)(n * nseg)));
    TRY(vals_alt.reserve((size_t)(n * nseg)));
    const u32 nblocks = (u32)div_up(n, RS_TILE);
    DevBuf<u32> hist, tmp;
    TRY(hist.alloc((size_t)256 * nblocks * (size_t)nseg));
    int passes = (key_bits + 7) / 8;
    if (passes < 1) passes = 1;
    for (int p = 0; p < passes; ++p) {
        const int shift = first_bit + 8 * p;
        hipLaunchKernelGGL(radix_hist, dim3(nblocks, (unsigned)nseg), dim3(RS_THREADS), 0, ctx->stream, keys.p, n, shift, hist.p, nblocks);
        TRY(chip_exclusive_scan_u32(ctx, hist.p, hist.p, (i64)256 * nblocks * nseg, tmp));
        hipLaunchKernelGGL(radix_scatter, dim3(nblocks, (unsigned)nseg), dim3(RS_THREADS), 0, ctx->stream, keys.p, vals.p,
                           keys_alt.p, vals_alt.p, n, shift, hist.p, nblocks);
        keys.swap(keys_alt);
        vals.swap(vals_alt);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
