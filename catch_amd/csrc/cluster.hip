// Sequence clustering pre-step: MinHash signatures and signature distances.
//
// Replaces, for `--cluster-and-design-separately` (catch/utils/cluster.py:358-430):
//   * lsh.MinHashFamily(k, N).make_h() / h(s)   (catch/utils/lsh.py:75-153): the
//     signature of a sequence is the N smallest values, kept with multiplicity
//     and sorted, of (a * md5(kmer) + b) mod (2^31 - 1) over every k-mer of the
//     sequence (md5 read as a 128-bit big-endian integer, :106-111); a sequence
//     with fewer than N k-mers repeats its k-mers in whole rounds (:126-135);
//   * MinHashFamily.estimate_jaccard_dist (:170-215): a merge walk of two
//     sorted signatures that stops after N union steps and counts the common
//     values -- since both signatures hold exactly N values the walk always
//     makes exactly N steps, so the distance is 1 - common / N.
//
// Kernels (all HBM- or integer-issue-bound, nothing GEMM-shaped):
//   kmer_md5_kernel    one thread per k-mer start: a 1024-position tile of the
//                      raw characters is staged in LDS, the single-block MD5 of
//                      the k <= 55 characters runs in registers, the 31-bit hash
//                      is written to H[position].  Positions whose k-mer would
//                      cross a sequence end hold garbage that nobody reads.
//   sig_select_kernel  one workgroup per sequence: a three-level radix select
//                      (11 + 10 + 10 bits, LDS histograms) finds the N-th
//                      smallest hash, the values below it are gathered, padded
//                      with copies of it, rank-sorted in LDS and written out.
//   sig_transpose_kernel  [seq][N] -> [N][seq] so the walks below read coalesced.
//   sig_row_kernel     one thread per sequence: walk against signature j (LDS).
//   sig_pairs_kernel   a T x T tile of pairs per workgroup, both signature
//                      tiles in LDS; writes float32 distances at the condensed
//                      index SciPy expects (cluster.py:86-100).
#include <algorithm>
#include <type_traits>

#include "internal.h"

struct catchhip_sigs {
    catchhip_ctx *ctx = nullptr;
    u32 nseq = 0, N = 0;
    DevBuf<u32> sig;    // [nseq][N], ascending
    DevBuf<u32> sigT;   // [N][nseq]
    DevBuf<u64> fpT;    // 4,096-bit fingerprints of the signatures, [word][nseq] (catchhip_sigs_neighbors_many, on first use)
    DevBuf<u32> fp_excess;
    bool fp_ready = false;
    // the neighbour graph (catchhip_sigs_graph): rows of g_idx / g_com delimited by g_ptr
    DevBuf<u64> g_ptr;
    DevBuf<u32> g_idx, g_com;
    u64 g_edges = 0;
    bool g_ready = false;
};

static int sigs_fingerprints(catchhip_ctx *ctx, catchhip_sigs *Sm, PhaseTimer &tm);

#define MD5_P 0x7FFFFFFFu
#define KM_THREADS 256
#define KM_PPT 4
#define KM_TILE (KM_THREADS * KM_PPT)
#define KM_MAXK 55
#define SS_THREADS 256
#define SS_MAXN 1024

__constant__ u32 c_md5_k[64];
__device__ __forceinline__ u32 rotl32(u32 x, int s) { return (x << s) | (x >> (32 - s)); }

// MD5 (RFC 1321) of one 64-byte block held as 16 little-endian words; returns
// the digest read as a big-endian 128-bit integer, reduced mod 2^31 - 1
// (2^32 = 2, 2^64 = 4, 2^96 = 8 mod 2^31 - 1).
__device__ __forceinline__ u32 md5_block_mod_p(const u32 (&M)[16]) {
    u32 a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
#define MD5_STEP(f, g, i, s)                                                  \
    {                                                                         \
        const u32 t = (f) + a + c_md5_k[i] + M[g];                            \
        a = d; d = c; c = b; b = b + rotl32(t, s);                            \
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int s = (i & 3) == 0 ? 7 : (i & 3) == 1 ? 12 : (i & 3) == 2 ? 17 : 22;
        MD5_STEP((b & c) | (~b & d), i, i, s);
    }
#pragma unroll
    for (int i = 16; i < 32; ++i) {
        const int s = (i & 3) == 0 ? 5 : (i & 3) == 1 ? 9 : (i & 3) == 2 ? 14 : 20;
        MD5_STEP((d & b) | (~d & c), (5 * i + 1) & 15, i, s);
    }
#pragma unroll
    for (int i = 32; i < 48; ++i) {
        const int s = (i & 3) == 0 ? 4 : (i & 3) == 1 ? 11 : (i & 3) == 2 ? 16 : 23;
        MD5_STEP(b ^ c ^ d, (3 * i + 5) & 15, i, s);
    }
#pragma unroll
    for (int i = 48; i < 64; ++i) {
        const int s = (i & 3) == 0 ? 6 : (i & 3) == 1 ? 10 : (i & 3) == 2 ? 15 : 21;
        MD5_STEP(c ^ (b | ~d), (7 * i) & 15, i, s);
    }
#undef MD5_STEP
    a += 0x67452301u; b += 0xefcdab89u; c += 0x98badcfeu; d += 0x10325476u;
    const u64 wa = __builtin_bswap32(a), wb = __builtin_bswap32(b), wc = __builtin_bswap32(c),
              wd = __builtin_bswap32(d);
    return mod_mersenne31(8 * wa + 4 * wb + 2 * wc + wd);
}

__global__ __launch_bounds__(KM_THREADS) void kmer_md5_kernel(const u32 *__restrict__ words, u64 total, int k,
                                                              u32 am, u32 b, u32 *__restrict__ H) {
    __shared__ u32 s_w[KM_TILE / 4 + 16];
    const u64 tile0 = (u64)blockIdx.x * KM_TILE;
    // the buffer is padded past `total`, so whole tiles (+ 64 bytes) can be read
    for (int w = threadIdx.x; w < KM_TILE / 4 + 16; w += KM_THREADS) s_w[w] = words[tile0 / 4 + w];
    __syncthreads();
    const int kfull = k >> 2, rem = k & 3;
#pragma unroll
    for (int q = 0; q < KM_PPT; ++q) {
        // consecutive lanes take consecutive positions (coalesced H writes)
        const int local = q * KM_THREADS + threadIdx.x;
        const u64 g = tile0 + local;
        if (g >= total) continue;
        const int w0 = local >> 2, sh = (local & 3) * 8;
        u32 M[16];
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            u32 v = 0;
            if (t <= kfull) {
                const u32 lo = s_w[w0 + t], hi = s_w[w0 + t + 1];
                v = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
                if (t == kfull) v = (v & ((1u << (8 * rem)) - 1u)) | (0x80u << (8 * rem));
            }
            M[t] = v;
        }
        M[14] = (u32)k * 8u;
        M[15] = 0;
        const u64 x = md5_block_mod_p(M);
        H[g] = mod_mersenne31((u64)am * x + b);
    }
}

// block-wide: which of the nbins weighted histogram bins holds the element of
// (1-based) rank `need`; returns the bin and the weighted count before it
__device__ __forceinline__ void select_bin(const u32 *hist, int nbins, u32 reps, u32 need, u32 *s_part, u32 *s_res) {
    const int per = nbins / SS_THREADS;   // 8 or 4
    u32 mine = 0;
    for (int t = 0; t < per; ++t) mine += hist[threadIdx.x * per + t] * reps;
    s_part[threadIdx.x] = mine;
    __syncthreads();
    // inclusive scan of 256 partial sums (Hillis-Steele)
    for (int off = 1; off < SS_THREADS; off <<= 1) {
        const u32 v = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    const u32 incl = s_part[threadIdx.x], excl = incl - mine;
    if (need > excl && need <= incl) {
        u32 run = excl;
        for (int t = 0; t < per; ++t) {
            const u32 c = hist[threadIdx.x * per + t] * reps;
            if (need <= run + c) {
                s_res[0] = (u32)(threadIdx.x * per + t);
                s_res[1] = run;
                break;
            }
            run += c;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(SS_THREADS) void sig_select_kernel(const u32 *__restrict__ H, const u64 *__restrict__ off,
                                                                u32 nseq, int k, u32 N, u32 *__restrict__ sig) {
    __shared__ u32 s_hist[2048];
    __shared__ u32 s_part[SS_THREADS];
    __shared__ u32 s_res[2];
    __shared__ u32 s_list[SS_MAXN];
    __shared__ u32 s_cnt;
    for (u32 s = blockIdx.x; s < nseq; s += gridDim.x) {
        const u64 base = off[s];
        const u64 nk = off[s + 1] - base - (u64)k + 1;
        // fewer k-mers than N: whole extra rounds over the k-mers (lsh.py:126-135)
        const u32 reps = nk >= N ? 1u : (u32)((N + nk - 1) / nk);
        u32 need = N, prefix = 0, pmask = 0, less = 0;
        for (int level = 0; level < 3; ++level) {
            const int shift = level == 0 ? 20 : level == 1 ? 10 : 0;
            const int nbins = level == 0 ? 2048 : 1024;
            for (int t = threadIdx.x; t < nbins; t += SS_THREADS) s_hist[t] = 0;
            __syncthreads();
            for (u64 i = threadIdx.x; i < nk; i += SS_THREADS) {
                const u32 v = H[base + i];
                if ((v & pmask) == prefix) atomicAdd(&s_hist[(v >> shift) & (nbins - 1)], 1u);
            }
            __syncthreads();
            select_bin(s_hist, nbins, reps, need, s_part, s_res);
            need -= s_res[1];
            less += s_res[1];
            prefix |= s_res[0] << shift;
            pmask |= (u32)(nbins - 1) << shift;
            __syncthreads();
        }
        // prefix is the N-th smallest value; `less` (< N) weighted values lie below it
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        for (u64 i = threadIdx.x; i < nk; i += SS_THREADS) {
            const u32 v = H[base + i];
            if (v < prefix) {
                const u32 at = atomicAdd(&s_cnt, reps);
                for (u32 r = 0; r < reps; ++r) s_list[at + r] = v;
            }
        }
        for (u32 t = less + threadIdx.x; t < N; t += SS_THREADS) s_list[t] = prefix;
        __syncthreads();
        for (u32 t = threadIdx.x; t < N; t += SS_THREADS) {
            const u32 v = s_list[t];
            u32 r = 0;
            for (u32 u = 0; u < N; ++u) {
                const u32 w = s_list[u];
                r += (w < v || (w == v && u < t)) ? 1u : 0u;
            }
            sig[(size_t)s * N + r] = v;
        }
        __syncthreads();
    }
}

__global__ void sig_transpose_kernel(const u32 *__restrict__ sig, u32 nseq, u32 N, u32 *__restrict__ sigT) {
    __shared__ u32 tile[32][33];
    const u32 s0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int y = threadIdx.y; y < 32; y += blockDim.y) {
        const u32 s = s0 + y, r = r0 + threadIdx.x;
        tile[y][threadIdx.x] = (s < nseq && r < N) ? sig[(size_t)s * N + r] : 0;
    }
    __syncthreads();
    for (int y = threadIdx.y; y < 32; y += blockDim.y) {
        const u32 r = r0 + y, s = s0 + threadIdx.x;
        if (s < nseq && r < N) sigT[(size_t)r * nseq + s] = tile[threadIdx.x][y];
    }
}

// common values met by the N-step merge walk of two ascending N-value lists.
// Branch-free: the three-way comparison of a divergent wavefront would run all
// three paths every step (measured: 28-33 us per row of 6,000 walks, one
// wavefront per CU, whether the values came from L2 or from LDS).
template <typename FA, typename FB> __device__ __forceinline__ u32 walk_common(u32 N, FA a_at, FB b_at) {
    u32 ia = 0, ib = 0, common = 0;
    for (u32 step = 0; step < N; ++step) {
        const u32 act = (ia < N) & (ib < N);           // the reference's loop condition
        const u32 a = a_at(min(ia, N - 1)), b = b_at(min(ib, N - 1));
        const u32 lt = act & (a < b), gt = act & (a > b), eq = act & (a == b);
        ia += lt | eq;
        ib += gt | eq;
        common += eq;
    }
    return common;
}

__global__ __launch_bounds__(256) void sig_row_kernel(const u32 *__restrict__ sig, const u32 *__restrict__ sigT,
                                                      u32 nseq, u32 N, u32 j, uint16_t *__restrict__ common) {
    extern __shared__ u32 s_a[];
    for (u32 t = threadIdx.x; t < N; t += blockDim.x) s_a[t] = sig[(size_t)j * N + t];
    __syncthreads();
    const u32 kq = blockIdx.x * blockDim.x + threadIdx.x;
    if (kq >= nseq) return;
    common[kq] = (uint16_t)walk_common(
        N, [&](u32 i) { return s_a[i]; }, [&](u32 i) { return sigT[(size_t)i * nseq + kq]; });
}

// The same with every thread's own signature staged in LDS first: the walk is a
// chain of ~N dependent reads, 0.3 us each from L2 (28 us per row at N = 100)
// against tens of ns from LDS; the staging loads are independent and coalesced
// ([value][sequence] layout).  64 threads per workgroup, 64 * N * 4 bytes of LDS.
__global__ __launch_bounds__(64) void sig_row_lds_kernel(const u32 *__restrict__ sig, const u32 *__restrict__ sigT,
                                                         u32 nseq, u32 N, u32 j, uint16_t *__restrict__ common) {
    extern __shared__ u32 s_mem[];
    u32 *s_a = s_mem, *s_b = s_mem + N;   // s_b[i * 64 + lane]
    for (u32 t = threadIdx.x; t < N; t += 64) s_a[t] = sig[(size_t)j * N + t];
    const u32 kq = blockIdx.x * 64 + threadIdx.x;
    const u32 kc = kq < nseq ? kq : nseq - 1;
    for (u32 i = 0; i < N; ++i) s_b[i * 64 + threadIdx.x] = sigT[(size_t)i * nseq + kc];
    __syncthreads();
    if (kq >= nseq) return;
    common[kq] = (uint16_t)walk_common(
        N, [&](u32 i) { return s_a[i]; }, [&](u32 i) { return s_b[i * 64 + threadIdx.x]; });
}

// The same row, reduced on the device to what the connected-components search looks at: the
// sequences whose walk against j finds at least min_common values (distance within the threshold), as
// (index << 16 | common) in arrival order (the host sorts the few of them)
__global__ __launch_bounds__(64) void sig_neigh_lds_kernel(const u32 *__restrict__ sig, const u32 *__restrict__ sigT,
                                                           u32 nseq, u32 N, u32 j, u32 min_common,
                                                           unsigned long long *__restrict__ out, u32 cap,
                                                           u32 *__restrict__ count) {
    extern __shared__ u32 s_mem[];
    u32 *s_a = s_mem, *s_b = s_mem + N;   // s_b[i * 64 + lane]
    for (u32 t = threadIdx.x; t < N; t += 64) s_a[t] = sig[(size_t)j * N + t];
    const u32 kq = blockIdx.x * 64 + threadIdx.x;
    const u32 kc = kq < nseq ? kq : nseq - 1;
    for (u32 i = 0; i < N; ++i) s_b[i * 64 + threadIdx.x] = sigT[(size_t)i * nseq + kc];
    __syncthreads();
    u32 c = 0;
    if (kq < nseq)
        c = walk_common(N, [&](u32 i) { return s_a[i]; }, [&](u32 i) { return s_b[i * 64 + threadIdx.x]; });
    const bool hit = kq < nseq && c >= min_common;
    const unsigned long long b = __ballot(hit);
    if (!b) return;
    u32 base = 0;
    if (threadIdx.x == 0) base = atomicAdd(count, (u32)__popcll(b));
    base = __shfl(base, 0, WAVE);
    if (hit) {
        const u32 pos = base + (u32)__popcll(b & ((1ull << threadIdx.x) - 1ull));
        if (pos < cap) out[pos] = ((unsigned long long)kq << 16) | (unsigned long long)c;
    }
}

// The neighbour lists of several vertices in one launch (the search asks for the vertex it explores and for
// those it is about to: the far neighbours it has just stacked).  Entry = query << 48 | index << 16 | common.
// Almost every (query, target) pair is a pair of unrelated sequences, and finding that out by the walk costs
// ~N dependent reads.  A fingerprint settles it first: every signature value sets one of 2,048 bits (32 words
// per sequence, stored [word][sequence]); `excess` = N - the bits set (values that fell on a bit already set,
// repeated values included).  The walk matches equal values pairwise, so
//     common <= sum over values of min(multiplicity in A, in B) <= popcount(bits A & bits B) + min(excess A, excess B)
// and a pair whose bound is below min_common cannot be a neighbour.  Unrelated signatures share two or three
// bits by chance; only the workgroups that hold a survivor stage their 64 signatures in LDS and walk.
#define NEIGH_MAXQ 32
// (Round 5: 2,048 bits.  With 4,096 the AND-popcount bound was ~250 of sig_graph_kernel's 419 ms at 182 VGPRs; 2,048 bits pass
// more related-but-not-near pairs to the walk, which is cheap since the survivors are walked lane-packed: 330 ms.  1,024 bits:
// 1,119 ms -- then nearly every tile of 32 queries holds a survivor and stages its signatures.)
#define NEIGH_FPW 32          // 64-bit words of a fingerprint
__device__ __forceinline__ u32 neigh_fp_bit(u32 v) { return (v * 2654435761u) >> 21; }   // 11 bits

__global__ __launch_bounds__(256) void sig_fp_build_kernel(const u32 *__restrict__ sig, u32 nseq, u32 N,
                                                           unsigned long long *__restrict__ fpT) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (u64)nseq * N) return;
    const u32 seq = (u32)(t / N);
    const u32 bit = neigh_fp_bit(sig[t]);
    atomicOr(&fpT[(size_t)(bit >> 6) * nseq + seq], 1ull << (bit & 63u));
}

__global__ __launch_bounds__(256) void sig_fp_excess_kernel(const unsigned long long *__restrict__ fpT, u32 nseq, u32 N,
                                                            u32 *__restrict__ excess) {
    const u32 seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    u32 set = 0;
    for (u32 w = 0; w < NEIGH_FPW; ++w) set += (u32)__popcll(fpT[(size_t)w * nseq + seq]);
    excess[seq] = N - set;
}

// 4 wavefronts per workgroup share its 64 targets: wavefront w takes the queries w, w + 4, ... (a target of a
// large cluster is a neighbour of every query of a call, and the walks of one lane are a chain)
#define NEIGH_WAVES 4
__global__ __launch_bounds__(64 * NEIGH_WAVES) void sig_neigh_many_kernel(
    const u32 *__restrict__ sig, const u32 *__restrict__ sigT, const unsigned long long *__restrict__ fpT,
    const u32 *__restrict__ excess, u32 nseq, u32 N, const u32 *__restrict__ js, u32 nq, u32 min_common,
    unsigned long long *__restrict__ out, u32 cap, u32 *__restrict__ count) {
    extern __shared__ u32 s_mem[];
    u32 *s_b = s_mem, *s_q = s_mem + (size_t)64 * N;   // s_b[i * 64 + lane], s_q[q * N + i]
    __shared__ unsigned long long s_qfp[NEIGH_MAXQ][NEIGH_FPW];
    __shared__ u32 s_qx[NEIGH_MAXQ];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 kq = blockIdx.x * 64 + lane;
    const u32 kc = kq < nseq ? kq : nseq - 1;
    for (u32 t = threadIdx.x; t < nq * NEIGH_FPW; t += 64 * NEIGH_WAVES)
        s_qfp[t / NEIGH_FPW][t % NEIGH_FPW] = fpT[(size_t)(t % NEIGH_FPW) * nseq + js[t / NEIGH_FPW]];
    if (threadIdx.x < nq) s_qx[threadIdx.x] = excess[js[threadIdx.x]];
    unsigned long long fp[NEIGH_FPW];
#pragma unroll
    for (int w = 0; w < NEIGH_FPW; ++w) fp[w] = fpT[(size_t)w * nseq + kc];
    const u32 ex = excess[kc];
    __syncthreads();
    u32 need = 0;                                      // the queries (of this wavefront) this target may be a neighbour of
    for (u32 q = wave; q < nq; q += NEIGH_WAVES) {
        u32 pc = 0;
#pragma unroll
        for (int w = 0; w < NEIGH_FPW; ++w) pc += (u32)__popcll(fp[w] & s_qfp[q][w]);
        if (kq < nseq && pc + min(ex, s_qx[q]) >= min_common) need |= 1u << q;
    }
    if (!__syncthreads_or(need != 0)) return;
    for (u32 t = threadIdx.x; t < 64 * N; t += 64 * NEIGH_WAVES) {
        const u32 i = t >> 6, l = t & 63;
        const u32 kk = blockIdx.x * 64 + l;
        s_b[t] = sigT[(size_t)i * nseq + (kk < nseq ? kk : nseq - 1)];
    }
    for (u32 t = threadIdx.x; t < nq * N; t += 64 * NEIGH_WAVES) s_q[t] = sig[(size_t)js[t / N] * N + (t % N)];
    __syncthreads();
    for (u32 q = wave; q < nq; q += NEIGH_WAVES) {
        if (!__ballot((need >> q) & 1u)) continue;
        const u32 *a = s_q + (size_t)q * N;
        u32 c = 0;
        if ((need >> q) & 1u)
            c = walk_common(N, [&](u32 i) { return a[i]; }, [&](u32 i) { return s_b[i * 64 + lane]; });
        const bool hit = c >= min_common && ((need >> q) & 1u);
        const unsigned long long b = __ballot(hit);
        if (!b) continue;
        u32 base = 0;
        if (lane == 0) base = atomicAdd(count, (u32)__popcll(b));
        base = __shfl(base, 0, WAVE);
        if (hit) {
            const u32 pos = base + (u32)__popcll(b & ((1ull << lane) - 1ull));
            if (pos < cap) out[pos] = ((unsigned long long)q << 48) | ((unsigned long long)kq << 16) | (unsigned long long)c;
        }
    }
}

// The WHOLE neighbour graph in one launch (round 4): every pair (q, t), q < t, whose walk finds at least min_common
// values, emitted in both directions as key = q << 32 | t, value = common.  The search then reads its lists from a
// CSR copy on the host instead of asking the device 11,000 times (S5 x 1.0: 224 k vertices; the calls were 3.3 of the
// 7 s of the search, and they compared 7.9e10 pairs where the triangle has 2.5e10).  Same two stages as above: the
// fingerprint bound settles almost every pair; workgroups that hold survivors stage signatures in LDS and walk.
// Workgroup (x, y): the 64 targets of block x against the SG_TILES tiles of SG_QT queries of group y.
#define SG_QT 32
#define SG_TILES 16
__global__ __launch_bounds__(64 * NEIGH_WAVES) void sig_graph_kernel(
    const u32 *__restrict__ sig, const u32 *__restrict__ sigT, const unsigned long long *__restrict__ fpT,
    const u32 *__restrict__ excess, u32 nseq, u32 N, u32 min_common,
    unsigned long long *__restrict__ out_key, u32 *__restrict__ out_val, unsigned long long cap,
    unsigned long long *__restrict__ count) {
    const u32 t0 = blockIdx.x * 64, q00 = blockIdx.y * (SG_QT * SG_TILES);
    if (q00 >= t0 + 63 || q00 >= nseq) return;             // only pairs with q < t
    extern __shared__ u32 s_mem[];
    u32 *s_b = s_mem, *s_q = s_mem + (size_t)64 * N;       // s_b[i * 64 + lane], s_q[q * N + i]
    __shared__ unsigned long long s_qfp[SG_QT][NEIGH_FPW + 1];
    __shared__ u32 s_qx[SG_QT], s_np;
    __shared__ uint16_t s_pairs[SG_QT * 64];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 kq = t0 + lane;
    const u32 kc = kq < nseq ? kq : nseq - 1;
    unsigned long long fp[NEIGH_FPW];
#pragma unroll
    for (int w = 0; w < NEIGH_FPW; ++w) fp[w] = fpT[(size_t)w * nseq + kc];
    const u32 ex = excess[kc];
    bool staged = false;
    for (u32 tile = 0; tile < SG_TILES; ++tile) {
        const u32 q0 = q00 + tile * SG_QT;
        if (q0 >= t0 + 63 || q0 >= nseq) break;            // (uniform)
        __syncthreads();                                    // the previous tile's walks are done with s_qfp / s_q
        for (u32 t = threadIdx.x; t < SG_QT * NEIGH_FPW; t += 64 * NEIGH_WAVES) {
            const u32 q = t % SG_QT, w = t / SG_QT;
            s_qfp[q][w] = fpT[(size_t)w * nseq + min(q0 + q, nseq - 1)];
        }
        if (threadIdx.x < SG_QT) s_qx[threadIdx.x] = excess[min(q0 + threadIdx.x, nseq - 1)];
        if (threadIdx.x == 0) s_np = 0;
        __syncthreads();
        u32 need = 0;
        for (u32 q = wave; q < SG_QT; q += NEIGH_WAVES) {
            u32 pc = 0;
#pragma unroll
            for (int w = 0; w < NEIGH_FPW; ++w) pc += (u32)__popcll(fp[w] & s_qfp[q][w]);
            if (kq < nseq && q0 + q < kq && pc + min(ex, s_qx[q]) >= min_common) need |= 1u << q;
        }
        if (!__syncthreads_or(need != 0)) continue;
        if (!staged) {
            for (u32 t = threadIdx.x; t < 64 * N; t += 64 * NEIGH_WAVES) {
                const u32 i = t >> 6, l = t & 63;
                s_b[t] = sigT[(size_t)i * nseq + min(t0 + l, nseq - 1)];
            }
            staged = true;
        }
        for (u32 t = threadIdx.x; t < SG_QT * N; t += 64 * NEIGH_WAVES) s_q[t] = sig[(size_t)min(q0 + t / N, nseq - 1) * N + (t % N)];
        __syncthreads();
        // Round 5: the survivors are walked LANE-PACKED.  A query's near targets are the same stretch of other strains'
        // genomes -- one or two of a wavefront's 64 consecutive targets --, so a wavefront that walked "its" query for the
        // lanes that needed it ran the ~N-step walk for one or two lanes at a time; the (query, target) pairs of the tile
        // are listed in LDS instead and every lane of the workgroup takes one.
        for (u32 q = wave; q < SG_QT; q += NEIGH_WAVES) {
            const bool mine = (need >> q) & 1u;
            const unsigned long long b = __ballot(mine);
            if (!b) continue;
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&s_np, (u32)__popcll(b));
            base = __shfl(base, 0, WAVE);
            if (mine) s_pairs[base + (u32)__popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)((q << 8) | lane);
        }
        __syncthreads();
        const u32 np = s_np;
        for (u32 p0 = wave * 64; p0 < np; p0 += 64 * NEIGH_WAVES) {
            const u32 p = p0 + lane;
            const bool have = p < np;
            const u32 pr = have ? s_pairs[p] : 0u, q = pr >> 8, tl = pr & 63u;
            const u32 *a = s_q + (size_t)q * N;
            u32 c = 0;
            if (have) c = walk_common(N, [&](u32 i) { return a[i]; }, [&](u32 i) { return s_b[i * 64 + tl]; });
            const bool hit = have && c >= min_common;
            const unsigned long long b = __ballot(hit);
            if (!b) continue;
            const u32 nh = (u32)__popcll(b);
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(count, 2ull * nh);
            base = __shfl(base, 0, WAVE);
            if (hit) {
                const unsigned long long pos = base + 2ull * (u32)__popcll(b & ((1ull << lane) - 1ull));
                const u32 kt = t0 + tl;
                if (pos + 1 < cap) {
                    out_key[pos] = ((unsigned long long)(q0 + q) << 32) | kt; out_val[pos] = c;
                    out_key[pos + 1] = ((unsigned long long)kt << 32) | (q0 + q); out_val[pos + 1] = c;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void sig_graph_ptr_kernel(const unsigned long long *__restrict__ key, u64 nedges, u32 nseq,
                                                            unsigned long long *__restrict__ ptr) {
    const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > nseq) return;
    const unsigned long long want = (unsigned long long)v << 32;
    u64 lo = 0, hi = nedges;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (key[mid] < want) lo = mid + 1; else hi = mid;
    }
    ptr[v] = lo;
}

__global__ __launch_bounds__(256) void sig_graph_idx_kernel(const unsigned long long *__restrict__ key, u64 nedges,
                                                            u32 *__restrict__ idx) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nedges) idx[e] = (u32)key[e];
}

__global__ __launch_bounds__(256) void sig_pairs_kernel(const u32 *__restrict__ sig, u32 nseq, u32 N, u32 T,
                                                        const float *__restrict__ lut, float *__restrict__ out) {
    extern __shared__ u32 s_ab[];
    const u32 bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;   // upper triangle of tiles only
    u32 *s_i = s_ab, *s_j = s_ab + (size_t)T * N;
    const u32 i0 = bi * T, j0 = bj * T;
    for (u32 t = threadIdx.x; t < T * N; t += blockDim.x) {
        const u32 r = t / N, c = t - r * N;
        s_i[t] = (i0 + r < nseq) ? sig[(size_t)(i0 + r) * N + c] : 0;
        s_j[t] = (j0 + r < nseq) ? sig[(size_t)(j0 + r) * N + c] : 0;
    }
    __syncthreads();
    for (u32 t = threadIdx.x; t < T * T; t += blockDim.x) {
        const u32 li = t / T, lj = t - li * T;
        const u64 i = i0 + li, j = j0 + lj;
        if (i >= j || j >= nseq) continue;
        const u32 *pa = s_i + (size_t)li * N, *pb = s_j + (size_t)lj * N;
        const u32 common = walk_common(
            N, [&](u32 x) { return pa[x]; }, [&](u32 x) { return pb[x]; });
        // condensed index of (i, j), i < j (cluster.py:94-99)
        const u64 idx = i * (u64)nseq - i * (i + 1) / 2 + (j - i - 1);
        out[idx] = lut[common];
    }
}

static bool g_md5_table_ready[64] = {};

static int md5_table_upload(catchhip_ctx *ctx) {
    if (ctx->device < 64 && g_md5_table_ready[ctx->device]) return 0;
    // RFC 1321 3.4: T[i] = floor(2^32 * |sin(i + 1)|), i in radians
    static const u32 T[64] = {
        0xd76aa478u, 0xe8c7b756u, 0x242070dbu, 0xc1bdceeeu, 0xf57c0fafu, 0x4787c62au, 0xa8304613u, 0xfd469501u,
        0x698098d8u, 0x8b44f7afu, 0xffff5bb1u, 0x895cd7beu, 0x6b901122u, 0xfd987193u, 0xa679438eu, 0x49b40821u,
        0xf61e2562u, 0xc040b340u, 0x265e5a51u, 0xe9b6c7aau, 0xd62f105du, 0x02441453u, 0xd8a1e681u, 0xe7d3fbc8u,
        0x21e1cde6u, 0xc33707d6u, 0xf4d50d87u, 0x455a14edu, 0xa9e3e905u, 0xfcefa3f8u, 0x676f02d9u, 0x8d2a4c8au,
        0xfffa3942u, 0x8771f681u, 0x6d9d6122u, 0xfde5380cu, 0xa4beea44u, 0x4bdecfa9u, 0xf6bb4b60u, 0xbebfbc70u,
        0x289b7ec6u, 0xeaa127fau, 0xd4ef3085u, 0x04881d05u, 0xd9d4d039u, 0xe6db99e5u, 0x1fa27cf8u, 0xc4ac5665u,
        0xf4292244u, 0x432aff97u, 0xab9423a7u, 0xfc93a039u, 0x655b59c3u, 0x8f0ccc92u, 0xffeff47du, 0x85845dd1u,
        0x6fa87e4fu, 0xfe2ce6e0u, 0xa3014314u, 0x4e0811a1u, 0xf7537e82u, 0xbd3af235u, 0x2ad7d2bbu, 0xeb86d391u};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_md5_k), T, sizeof(T)));
    if (ctx->device < 64) g_md5_table_ready[ctx->device] = true;
    return 0;
}

// upload(dst, stream): queues the copies of the concatenated characters to dst (null: one copy from `bytes`)
template <class Upload>
static int sigs_create_impl(catchhip_ctx *ctx, const u8 *bytes, const u64 *offsets, u32 nseq, i32 k, u32 N,
                            u32 a, u32 b, catchhip_sigs **out, Upload upload);
struct SigsNoUpload { int operator()(u8 *, hipStream_t) const { return 1; } };

extern "C" int catchhip_sigs_create(catchhip_ctx *ctx, const u8 *bytes, const u64 *offsets, u32 nseq, i32 k, u32 N,
                                    u32 a, u32 b, catchhip_sigs **out) {
    return sigs_create_impl(ctx, bytes, offsets, nseq, k, N, a, b, out, SigsNoUpload());
}

template <class Upload>
static int sigs_create_impl(catchhip_ctx *ctx, const u8 *bytes, const u64 *offsets, u32 nseq, i32 k, u32 N,
                            u32 a, u32 b, catchhip_sigs **out, Upload upload) {
    ARG_CHECK(ctx && out && k >= 1 && k <= KM_MAXK && N >= 1 && N <= SS_MAXN && nseq < (1u << 30));
    ARG_CHECK(a >= 1 && a <= MD5_P && b <= MD5_P);
    PoolScope pool_scope(ctx);
    *out = nullptr;
    catchhip_sigs *S = new catchhip_sigs();
    S->ctx = ctx;
    S->nseq = nseq;
    S->N = N;
    if (nseq == 0) { *out = S; return 0; }
    const bool own_upload = std::is_same<Upload, SigsNoUpload>::value;
    if (!((bytes || !own_upload) && offsets && offsets[0] == 0)) {
        delete S;
        ARG_CHECK((bytes || !own_upload) && offsets && offsets[0] == 0);
    }
    for (u32 s = 0; s < nseq; ++s) {
        // lsh.py:113 asserts kmer_size <= len(s)
        if (offsets[s + 1] < offsets[s] || offsets[s + 1] - offsets[s] < (u64)k) {
            delete S;
            chip_set_error("signatures: sequence %u is shorter than the k-mer size %d", s, (int)k);
            return CATCHHIP_EINVAL;
        }
    }
    const u64 total = offsets[nseq];
    int rc = 0;
    do {
        if ((rc = hipSetDevice(ctx->device) == hipSuccess ? 0 : CATCHHIP_EHIP)) break;
        if ((rc = md5_table_upload(ctx))) break;
        hipStream_t st = ctx->stream;
        const u64 ntiles = (total + KM_TILE - 1) / KM_TILE;
        DevBuf<u32> d_words, H;
        DevBuf<u64> d_off;
        const size_t padded = (size_t)ntiles * KM_TILE + 64;
        if ((rc = d_words.alloc(padded / 4)) || (rc = H.alloc((size_t)total)) || (rc = d_off.alloc((size_t)nseq + 1)) ||
            (rc = S->sig.alloc((size_t)nseq * N)) || (rc = S->sigT.alloc((size_t)nseq * N)))
            break;
#define CL_HIP(expr)                                                                        \
    if ((expr) != hipSuccess) {                                                             \
        chip_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(hipGetLastError())); \
        rc = CATCHHIP_EHIP;                                                                 \
        break;                                                                              \
    }
        CL_HIP(hipMemsetAsync((u8 *)d_words.p + (total & ~(u64)3), 0, padded - (total & ~(u64)3), st));
        if (own_upload) { CL_HIP(hipMemcpyAsync(d_words.p, bytes, total, hipMemcpyHostToDevice, st)); }
        else if ((rc = upload((u8 *)d_words.p, st))) break;
        CL_HIP(hipMemcpyAsync(d_off.p, offsets, sizeof(u64) * ((size_t)nseq + 1), hipMemcpyHostToDevice, st));
        PhaseTimer tm(ctx, PHASE_NDF);
        hipLaunchKernelGGL(kmer_md5_kernel, dim3((unsigned)ntiles), dim3(KM_THREADS), 0, st, (const u32 *)d_words.p,
                           total, (int)k, (u32)(a % MD5_P), b, H.p);
        hipLaunchKernelGGL(sig_select_kernel, dim3((unsigned)std::min<u64>(nseq, (u64)ctx->num_cus * 32)),
                           dim3(SS_THREADS), 0, st, (const u32 *)H.p, (const u64 *)d_off.p, nseq, (int)k, N, S->sig.p);
        hipLaunchKernelGGL(sig_transpose_kernel, dim3((nseq + 31) / 32, (N + 31) / 32), dim3(32, 8), 0, st,
                           (const u32 *)S->sig.p, nseq, N, S->sigT.p);
        tm.launch(3);
        CL_HIP(hipGetLastError());
        CL_HIP(hipStreamSynchronize(st));
        tm.finish();
    } while (0);
    if (rc) { delete S; return rc; }
    *out = S;
    return 0;
}

extern "C" void catchhip_sigs_destroy(catchhip_sigs *S) {
    if (!S) return;
    PoolScope pool_scope(S->ctx);
    delete S;
}

extern "C" int catchhip_sigs_fetch(catchhip_ctx *ctx, const catchhip_sigs *S, u32 *out) {
    ARG_CHECK(ctx && S && S->ctx == ctx);
    if (S->nseq == 0) return 0;
    ARG_CHECK(out);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(out, S->sig.p, sizeof(u32) * (size_t)S->nseq * S->N, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// the sequences as one pointer per string (their own storage): gathered by host threads into pinned memory
// instead of a multi-gigabyte join + encode in the interpreter (S5 x 1.0: 3.5 Gbases of fragments, 1.3 s)
#include <memory>
#include <new>
#include <thread>
extern "C" int catchhip_sigs_create_ptrs(catchhip_ctx *ctx, const u8 *const *seq_ptr, const i64 *seq_len, u32 nseq, i32 k,
                                         u32 N, u32 a, u32 b, catchhip_sigs **out) {
    ARG_CHECK(ctx && out && (nseq == 0 || (seq_ptr && seq_len)));
    std::vector<u64> off((size_t)nseq + 1, 0);
    for (u32 i = 0; i < nseq; ++i) {
        ARG_CHECK(seq_len[i] >= 0 && (seq_len[i] == 0 || seq_ptr[i] != nullptr));
        off[i + 1] = off[i] + (u64)seq_len[i];
    }
    const u64 total = off[nseq];
    // The characters go up in chunks through two pinned buffers: host threads gather chunk c while chunk c - 1 is on
    // its way (round 4; until then: gathered into one pageable array of the whole size and copied from there --
    // fresh pages for 3.5 GB and a staged copy, 0.8 of the 1.0 s the signatures of S5 x 1.0 took).
    const size_t CH = (size_t)64 << 20;
    auto upload = [&](u8 *dst, hipStream_t st) -> int {
        if (total == 0) return 0;
        TRY(chip_pinned_reserve(ctx, 2 * CH));
        u8 *pin = (u8 *)ctx->h_big;
        hipEvent_t ev[2] = {nullptr, nullptr};
        HIP_TRY(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
        int rc = 0;
        const int nthreads = (int)std::max<u64>(1, std::min<u64>(8, total >> 21));
        for (u64 c = 0, lo = 0; lo < total && !rc; ++c, lo += CH) {
            const u64 hi = std::min<u64>(total, lo + CH);
            u8 *buf = pin + (c & 1) * CH;
            if (c >= 2 && hipEventSynchronize(ev[c & 1]) != hipSuccess) { rc = CATCHHIP_EHIP; break; }
            auto work = [&](int tix) {
                const u64 a0 = lo + (hi - lo) * (u64)tix / (u64)nthreads, a1 = lo + (hi - lo) * (u64)(tix + 1) / (u64)nthreads;
                if (a0 >= a1) return;
                size_t i = (size_t)(std::upper_bound(off.begin(), off.end(), a0) - off.begin()) - 1;   // off[i] <= a0 < off[i + 1]
                for (u64 at = a0; at < a1 && i < nseq; ++i) {
                    const u64 e = std::min<u64>(off[i + 1], a1);
                    if (e > at) { memcpy(buf + (at - lo), seq_ptr[i] + (at - off[i]), (size_t)(e - at)); at = e; }
                }
            };
            if (nthreads == 1) work(0);
            else {
                std::vector<std::thread> th;
                for (int tix = 0; tix < nthreads; ++tix) th.emplace_back(work, tix);
                for (auto &t : th) t.join();
            }
            if (hipMemcpyAsync(dst + lo, buf, (size_t)(hi - lo), hipMemcpyHostToDevice, st) != hipSuccess ||
                hipEventRecord(ev[c & 1], st) != hipSuccess) rc = CATCHHIP_EHIP;
        }
        if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = CATCHHIP_EHIP;      // (the pinned buffers are reused by others)
        (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
        if (rc) chip_set_error("signatures: upload failed");
        return rc;
    };
    return sigs_create_impl(ctx, nullptr, off.data(), nseq, k, N, a, b, out, upload);
}

extern "C" int catchhip_sigs_common_row(catchhip_ctx *ctx, const catchhip_sigs *S, u32 j, uint16_t *common) {
    ARG_CHECK(ctx && S && S->ctx == ctx && common && j < S->nseq);
    PoolScope pool_scope(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<uint16_t> d;
    TRY(d.alloc(S->nseq));
    TRY(chip_pinned_reserve(ctx, sizeof(uint16_t) * (size_t)S->nseq));
    PhaseTimer tm(ctx, PHASE_NDF);
    if (S->N <= 176)   // 65 * N * 4 bytes of LDS <= 45 KB
        hipLaunchKernelGGL(sig_row_lds_kernel, dim3((S->nseq + 63) / 64), dim3(64), sizeof(u32) * 65 * (size_t)S->N, st,
                           (const u32 *)S->sig.p, (const u32 *)S->sigT.p, S->nseq, S->N, j, d.p);
    else
        hipLaunchKernelGGL(sig_row_kernel, dim3((S->nseq + 255) / 256), dim3(256), sizeof(u32) * S->N, st,
                           (const u32 *)S->sig.p, (const u32 *)S->sigT.p, S->nseq, S->N, j, d.p);
    tm.launch(1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ctx->h_big, d.p, sizeof(uint16_t) * (size_t)S->nseq, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    tm.finish();
    memcpy(common, ctx->h_big, sizeof(uint16_t) * (size_t)S->nseq);
    return 0;
}

extern "C" int catchhip_sigs_neighbors(catchhip_ctx *ctx, const catchhip_sigs *S, u32 j, u32 min_common,
                                       unsigned long long *out, i64 cap, i64 *count) {
    ARG_CHECK(ctx && S && S->ctx == ctx && out && count && cap >= 0 && j < S->nseq);
    ARG_CHECK(S->N <= 176);   // (the caller uses catchhip_sigs_common_row for longer signatures)
    PoolScope pool_scope(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const u32 dcap = S->nseq;
    DevBuf<unsigned long long> d;
    DevBuf<u32> d_n;
    TRY(d.alloc((size_t)dcap + 1));
    TRY(d_n.alloc(1));
    // the count and the first entries come back together; a longer list takes a second copy
    const size_t first = 4096;
    TRY(chip_pinned_reserve(ctx, sizeof(unsigned long long) * ((size_t)dcap + 2)));
    HIP_TRY(hipMemsetAsync(d_n.p, 0, sizeof(u32), st));
    PhaseTimer tm(ctx, PHASE_NDF);
    hipLaunchKernelGGL(sig_neigh_lds_kernel, dim3((S->nseq + 63) / 64), dim3(64), sizeof(u32) * 65 * (size_t)S->N, st,
                       (const u32 *)S->sig.p, (const u32 *)S->sigT.p, S->nseq, S->N, j, min_common, d.p, dcap, d_n.p);
    tm.launch(1);
    HIP_TRY(hipGetLastError());
    unsigned long long *h = (unsigned long long *)ctx->h_big;
    const size_t n0 = std::min<size_t>(first, dcap);
    HIP_TRY(hipMemcpyAsync(h, d_n.p, sizeof(u32), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h + 1, d.p, sizeof(unsigned long long) * n0, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const u32 n = *(volatile u32 *)h;
    if (n > n0) {
        HIP_TRY(hipMemcpyAsync(h + 1 + n0, d.p + n0, sizeof(unsigned long long) * (n - n0), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    tm.finish();
    *count = n;
    if ((i64)n > cap) { chip_set_error("sigs_neighbors: %u neighbours, room for %lld", n, (long long)cap); return CATCHHIP_EINVAL; }
    memcpy(out, h + 1, sizeof(unsigned long long) * (size_t)n);
    return 0;
}

extern "C" int catchhip_sigs_neighbors_many(catchhip_ctx *ctx, const catchhip_sigs *S, const u32 *js, i64 nq,
                                            u32 min_common, unsigned long long *out, i64 cap, i64 *count) {
    ARG_CHECK(ctx && S && S->ctx == ctx && js && out && count && cap >= 0 && nq >= 1 && nq <= NEIGH_MAXQ);
    ARG_CHECK(S->N <= 112);   // (64 + 32) * N * 4 bytes of LDS <= 43 KB
    for (i64 q = 0; q < nq; ++q) ARG_CHECK(js[q] < S->nseq);
    PoolScope pool_scope(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t dcap = (size_t)std::min<i64>(cap, (i64)S->nseq * nq);
    DevBuf<unsigned long long> d;
    DevBuf<u32> d_n, d_js;
    TRY(d.alloc(dcap + 1));
    TRY(d_n.alloc(1));
    TRY(d_js.alloc(NEIGH_MAXQ));
    const size_t first = 16384;
    TRY(chip_pinned_reserve(ctx, sizeof(unsigned long long) * (dcap + 2)));
    HIP_TRY(hipMemsetAsync(d_n.p, 0, sizeof(u32), st));
    HIP_TRY(hipMemcpyAsync(d_js.p, js, sizeof(u32) * (size_t)nq, hipMemcpyHostToDevice, st));
    PhaseTimer tm(ctx, PHASE_NDF);
    catchhip_sigs *Sm = const_cast<catchhip_sigs *>(S);
    TRY(sigs_fingerprints(ctx, Sm, tm));                // on first use
    hipLaunchKernelGGL(sig_neigh_many_kernel, dim3((S->nseq + 63) / 64), dim3(64 * NEIGH_WAVES), sizeof(u32) * (64 + (size_t)nq) * S->N, st,
                       (const u32 *)S->sig.p, (const u32 *)S->sigT.p, (const unsigned long long *)Sm->fpT.p,
                       (const u32 *)Sm->fp_excess.p, S->nseq, S->N, (const u32 *)d_js.p, (u32)nq,
                       min_common, d.p, (u32)std::min<size_t>(dcap, 0xffffffffu), d_n.p);
    tm.launch(1);
    HIP_TRY(hipGetLastError());
    unsigned long long *h = (unsigned long long *)ctx->h_big;
    const size_t n0 = std::min<size_t>(first, dcap);
    HIP_TRY(hipMemcpyAsync(h, d_n.p, sizeof(u32), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h + 1, d.p, sizeof(unsigned long long) * n0, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));   // (js is read by its copy until here)
    const u32 n = *(volatile u32 *)h;
    *count = n;
    if ((i64)n > cap) { tm.finish(); chip_set_error("sigs_neighbors_many: %u neighbours, room for %lld", n, (long long)cap); return CATCHHIP_EINVAL; }
    if (n > n0) {
        HIP_TRY(hipMemcpyAsync(h + 1 + n0, d.p + n0, sizeof(unsigned long long) * (n - n0), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    tm.finish();
    memcpy(out, h + 1, sizeof(unsigned long long) * (size_t)n);
    return 0;
}

static int sigs_fingerprints(catchhip_ctx *ctx, catchhip_sigs *Sm, PhaseTimer &tm) {
    hipStream_t st = ctx->stream;
    if (Sm->fp_ready) return 0;
    TRY(Sm->fpT.alloc((size_t)NEIGH_FPW * Sm->nseq));
    TRY(Sm->fp_excess.alloc(Sm->nseq));
    HIP_TRY(hipMemsetAsync(Sm->fpT.p, 0, sizeof(u64) * NEIGH_FPW * (size_t)Sm->nseq, st));
    hipLaunchKernelGGL(sig_fp_build_kernel, dim3((unsigned)(((u64)Sm->nseq * Sm->N + 255) / 256)), dim3(256), 0, st,
                       (const u32 *)Sm->sig.p, Sm->nseq, Sm->N, (unsigned long long *)Sm->fpT.p);
    hipLaunchKernelGGL(sig_fp_excess_kernel, dim3((Sm->nseq + 255) / 256), dim3(256), 0, st,
                       (const unsigned long long *)Sm->fpT.p, Sm->nseq, Sm->N, Sm->fp_excess.p);
    tm.launch(2);
    Sm->fp_ready = true;
    return 0;
}

static int ceil_log2_u32(u32 v) { int b = 0; while (((u64)1 << b) < (u64)v) ++b; return b; }

extern "C" int catchhip_sigs_graph(catchhip_ctx *ctx, catchhip_sigs *S, u32 min_common, i64 max_edges, i64 *nedges) {
    ARG_CHECK(ctx && S && S->ctx == ctx && nedges && max_edges >= 0);
    ARG_CHECK(S->N <= 112 && S->nseq >= 1 && S->nseq < (1u << 31));
    PoolScope pool_scope(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    PhaseTimer tm(ctx, PHASE_NDF);
    TRY(sigs_fingerprints(ctx, S, tm));
    S->g_ready = false;
    DevBuf<u64> key, key_alt, d_n;
    DevBuf<u32> val, val_alt;
    TRY(d_n.alloc(1));
    TRY(chip_pinned_reserve(ctx, 64));
    // (S5 x 1.0: 224 k vertices, 173 M ordered pairs, 771 per vertex; a second pass costs another 0.75 s)
    u64 cap = std::max<u64>((u64)1 << 24, (u64)1024 * S->nseq);
    if (max_edges && cap > (u64)max_edges) cap = (u64)max_edges;
    u64 n = 0;
    const dim3 grid((S->nseq + 63) / 64, (S->nseq + SG_QT * SG_TILES - 1) / (SG_QT * SG_TILES));
    ARG_CHECK(grid.y <= 65535);
    for (int attempt = 0; attempt < 2; ++attempt) {
        TRY(key.alloc(cap + 2));
        TRY(val.alloc(cap + 2));
        HIP_TRY(hipMemsetAsync(d_n.p, 0, sizeof(u64), st));
        hipLaunchKernelGGL(sig_graph_kernel, grid, dim3(64 * NEIGH_WAVES), sizeof(u32) * (64 + (size_t)SG_QT) * S->N, st,
                           (const u32 *)S->sig.p, (const u32 *)S->sigT.p, (const unsigned long long *)S->fpT.p,
                           (const u32 *)S->fp_excess.p, S->nseq, S->N, min_common, (unsigned long long *)key.p, val.p,
                           (unsigned long long)cap, (unsigned long long *)d_n.p);
        tm.launch(1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ctx->h_pin, d_n.p, sizeof(u64), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        n = *(volatile u64 *)ctx->h_pin;
        if (n + 1 < cap || n == 0) break;
        if (max_edges && n > (u64)max_edges) { tm.finish(); *nedges = (i64)n; return 0; }   // too many: the caller asks list by list
        if (attempt == 1) { chip_set_error("sigs_graph: edge count changed between two passes"); return CATCHHIP_EINVAL; }
        key.release(); val.release();
        cap = n + 2;
    }
    *nedges = (i64)n;
    if (n) {
        TRY(key_alt.alloc(cap + 2));
        TRY(val_alt.alloc(cap + 2));
        const int bits = std::max(1, ceil_log2_u32(S->nseq));
        TRY(chip_radix_sort_pairs(ctx, key, key_alt, val, val_alt, (i64)n, bits, 0));
        TRY(chip_radix_sort_pairs(ctx, key, key_alt, val, val_alt, (i64)n, bits, 32));
    }
    TRY(S->g_ptr.alloc((size_t)S->nseq + 1));
    TRY(S->g_idx.alloc(std::max<u64>(n, 1)));
    hipLaunchKernelGGL(sig_graph_ptr_kernel, dim3((S->nseq + 256) / 256), dim3(256), 0, st,
                       (const unsigned long long *)key.p, n, S->nseq, (unsigned long long *)S->g_ptr.p);
    if (n) hipLaunchKernelGGL(sig_graph_idx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                              (const unsigned long long *)key.p, n, S->g_idx.p);
    tm.launch(2);
    HIP_TRY(hipGetLastError());
    S->g_com.swap(val);
    HIP_TRY(hipStreamSynchronize(st));
    tm.finish();
    S->g_edges = n;
    S->g_ready = true;
    return 0;
}

extern "C" int catchhip_sigs_graph_fetch(catchhip_ctx *ctx, const catchhip_sigs *S, i64 *ptr, u32 *idx, u32 *common) {
    ARG_CHECK(ctx && S && S->ctx == ctx && ptr && S->g_ready && (S->g_edges == 0 || (idx && common)));
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ptr, S->g_ptr.p, sizeof(u64) * ((size_t)S->nseq + 1), hipMemcpyDeviceToHost, st));
    if (S->g_edges) {
        HIP_TRY(hipMemcpyAsync(idx, S->g_idx.p, sizeof(u32) * (size_t)S->g_edges, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(common, S->g_com.p, sizeof(u32) * (size_t)S->g_edges, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

extern "C" int catchhip_sigs_condensed(catchhip_ctx *ctx, const catchhip_sigs *S, const float *lut, float *out) {
    ARG_CHECK(ctx && S && S->ctx == ctx && lut);
    PoolScope pool_scope(ctx);
    const u64 n = S->nseq;
    if (n < 2) return 0;
    ARG_CHECK(out);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const u64 npairs = n * (n - 1) / 2;
    DevBuf<float> d_out, d_lut;
    TRY(d_out.alloc((size_t)npairs));
    TRY(d_lut.alloc((size_t)S->N + 1));
    HIP_TRY(hipMemcpyAsync(d_lut.p, lut, sizeof(float) * ((size_t)S->N + 1), hipMemcpyHostToDevice, st));
    // both signature tiles of a workgroup live in LDS (<= 48 KB)
    u32 T = 32;
    while (T > 4 && sizeof(u32) * 2 * (size_t)T * S->N > 48 * 1024) T >>= 1;
    const u32 nt = (u32)((n + T - 1) / T);
    ARG_CHECK(nt <= 65535);
    PhaseTimer tm(ctx, PHASE_NDF);
    hipLaunchKernelGGL(sig_pairs_kernel, dim3(nt, nt), dim3(256), sizeof(u32) * 2 * (size_t)T * S->N, st,
                       (const u32 *)S->sig.p, S->nseq, S->N, T, (const float *)d_lut.p, d_out.p);
    tm.launch(1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out.p, sizeof(float) * (size_t)npairs, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    tm.finish();
    return 0;
}
