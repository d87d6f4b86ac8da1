// Context, error handling, input upload and bit-plane packing.
#include <stdarg.h>

#include <algorithm>
#include <vector>

#include "internal.h"

static thread_local char g_err[1024] = "";

void chip_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- caching device allocator (see internal.h) ---------------------------
#include <map>
#include <mutex>
#include <unordered_map>

namespace {
std::mutex g_pool_mu;
// free blocks are cached per (owner, size class): the owner is the context the
// calling entry point works for (PoolScope), so a block never moves between
// streams -- reuse inside one context is ordered by its stream
std::multimap<std::pair<const void *, size_t>, void *> g_pool_free;
struct PoolLive { const void *owner; size_t cls; };
std::unordered_map<void *, PoolLive> g_pool_live;
thread_local const void *g_pool_owner = nullptr;
// [0] hipMalloc calls, [1] bytes obtained from the driver and still held,
// [2] bytes cached (free), [3] hipMalloc retries after freeing the cache
long long g_pool_stats[4] = {0, 0, 0, 0};

// Blocks above 64 MiB are few and large: an exact-class cache would keep one
// set per group size (S4: 20 groups, each with its own 64 MiB-granule sizes),
// so a request takes the smallest cached block of its owner that is large
// enough (at most twice the request); the request sequence of a step repeats,
// so the assignment settles after the first pass.
const size_t POOL_BIG = (size_t)1 << 26;

size_t pool_class(size_t b) {
    if (b < 512) return 512;
    if (b <= ((size_t)1 << 26)) {
        size_t c = 512;
        while (c < b) c <<= 1;
        return c;
    }
    // 64 MiB granules above 64 MiB; from 1 GiB an eighth of the power of two below the request (requests that
    // differ by a few per cent -- the seed work lists of the chunks of a union instance: 28.6, 28.8, 28.7 GB --
    // then share a class instead of each costing a hipFree + hipMalloc of tens of GB, seconds apiece)
    size_t step = (size_t)1 << 26;
    if (b >= ((size_t)1 << 30)) {
        size_t p2 = (size_t)1 << 30;
        while ((p2 << 1) <= b) p2 <<= 1;
        step = p2 >> 3;
    }
    return (b + step - 1) / step * step;
}
}  // namespace

// what the library may hold before a request that misses the cache returns idle blocks to the driver
static size_t pool_soft_limit() {
    // (initialised once, under call_once: lanes and the prefetch thread allocate concurrently; a fractional
    // CATCHHIP_POOL_SOFT_LIMIT_GB is scaled before it is truncated, and at least one byte so that "0.0001" is not
    // mistaken for "no limit")
    static std::once_flag once;
    static size_t soft_limit = 0;
    std::call_once(once, [] {
        size_t fr = 0, tot = 0;
        // (0.87 of the device since round 6, 0.72 until then: configs[4]'s two front-end workers want ~243 GB of blocks at
        // their peaks, and at 207 GB every step returned blocks to the driver and asked for them again -- 547 instead of ~400
        // hipMalloc calls in a three-step run and a step of 3.95-4.25 s instead of 3.3-3.45; a request that the driver cannot
        // serve still trims the cache and tries again, chip_pool_alloc below)
        soft_limit = (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) ? (size_t)((double)tot * 0.87) : ~(size_t)0 >> 1;
        if (const char *e = getenv("CATCHHIP_POOL_SOFT_LIMIT_GB")) {
            const double gb = atof(e);
            if (gb > 0.0) soft_limit = std::max<size_t>((size_t)(gb * (double)((size_t)1 << 30)), 1);
        }
    });
    return soft_limit;
}

const void *chip_pool_set_owner(const void *owner) {
    const void *prev = g_pool_owner;
    g_pool_owner = owner;
    return prev;
}

void *chip_pool_alloc(size_t bytes) {
    const void *owner = g_pool_owner;
    const size_t cls = pool_class(bytes);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        // exact size class only: a looser fit lets one request take the block the
        // next one needs and the steady state (no hipMalloc at all) is lost
        auto it = g_pool_free.find(std::make_pair(owner, cls));
        if (it == g_pool_free.end() && cls > POOL_BIG) {
            it = g_pool_free.lower_bound(std::make_pair(owner, cls));
            // (at most twice the request -- unless a new block would take the library over its soft limit:
            // then any idle block that is large enough beats returning the cache to the driver)
            if (it != g_pool_free.end() &&
                (it->first.first != owner ||
                 (it->first.second > 2 * cls && (size_t)g_pool_stats[1] + cls <= pool_soft_limit())))
                it = g_pool_free.end();
        }
        if (it != g_pool_free.end()) {
            void *p = it->second;
            const size_t got = it->first.second;
            g_pool_free.erase(it);
            g_pool_live[p] = PoolLive{owner, got};
            g_pool_stats[2] -= (long long)got;
            return p;
        }
    }
    // The cache keeps what it was given; a request it cannot serve while the library already holds most
    // of the device (a union instance of a clustered design: work lists of 100+ GB whose sizes differ from
    // chunk to chunk) first returns this owner's idle blocks to the driver, so that the runtime's own
    // allocations (kernel arguments, scratch) do not fail with the memory sitting unused in here.
    // (hipFree waits for the device, so blocks that queued work still reads are safe to return.)
    {
        const size_t soft_limit = pool_soft_limit();
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if ((size_t)g_pool_stats[1] + cls > soft_limit && g_pool_stats[2] > 0) {
            for (auto it = g_pool_free.begin(); it != g_pool_free.end();) {
                if (it->first.first == owner) {
                    (void)hipFree(it->second);
                    g_pool_stats[1] -= (long long)it->first.second;
                    g_pool_stats[2] -= (long long)it->first.second;
                    it = g_pool_free.erase(it);
                } else ++it;
            }
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, cls);
    g_pool_stats[0]++;
    if (e != hipSuccess) {
        g_pool_stats[3]++;
        // give this owner's cached blocks back to the driver and retry once
        // (other owners' blocks may still be referenced by queued work)
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            for (auto it = g_pool_free.begin(); it != g_pool_free.end();) {
                if (it->first.first == owner) {
                    (void)hipFree(it->second);
                    g_pool_stats[1] -= (long long)it->first.second;
                    g_pool_stats[2] -= (long long)it->first.second;
                    it = g_pool_free.erase(it);
                } else ++it;
            }
        }
        e = hipMalloc(&p, cls);
        if (e != hipSuccess) {
            chip_set_error("hipMalloc(%zu bytes) failed: %s", cls, hipGetErrorString(e));
            return nullptr;
        }
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_live[p] = PoolLive{owner, cls};
    g_pool_stats[1] += (long long)cls;
    return p;
}

void chip_pool_free(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_live.find(p);
    if (it == g_pool_live.end()) return;
    g_pool_free.insert(std::make_pair(std::make_pair(it->second.owner, it->second.cls), p));
    g_pool_stats[2] += (long long)it->second.cls;
    g_pool_live.erase(it);
}

// a context is going away: its cached blocks go back to the driver
void chip_pool_release_owner(const void *owner) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto it = g_pool_free.begin(); it != g_pool_free.end();) {
        if (it->first.first == owner) {
            (void)hipFree(it->second);
            g_pool_stats[1] -= (long long)it->first.second;
            g_pool_stats[2] -= (long long)it->first.second;
            it = g_pool_free.erase(it);
        } else ++it;
    }
}

// every idle block of every context back to the driver (after the device has finished: queued work may still
// read them) -- between phases of a long-running process whose next phase allocates differently
extern "C" int catchhip_pool_trim(void) {
    HIP_TRY(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto it = g_pool_free.begin(); it != g_pool_free.end(); it = g_pool_free.erase(it)) {
        (void)hipFree(it->second);
        g_pool_stats[1] -= (long long)it->first.second;
        g_pool_stats[2] -= (long long)it->first.second;
    }
    return 0;
}

extern "C" int catchhip_pool_stats(int64_t *out4) {
    ARG_CHECK(out4 != nullptr);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (int i = 0; i < 4; ++i) out4[i] = g_pool_stats[i];
    return 0;
}

extern "C" const char *catchhip_last_error(void) { return g_err; }
extern "C" int catchhip_abi_version(void) { return CATCHHIP_ABI_VERSION; }

extern "C" int catchhip_device_count(int *count) {
    ARG_CHECK(count != nullptr);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        chip_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return CATCHHIP_EHIP;
    }
    *count = n;
    return 0;
}

extern "C" int catchhip_ctx_create(int device, catchhip_ctx **out) {
    ARG_CHECK(out != nullptr);
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    catchhip_ctx *c = new catchhip_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        chip_set_error("hipStreamCreate failed");
        return CATCHHIP_EHIP;
    }
    for (int i = 0; i < 2 * NPHASE; ++i) (void)hipEventCreate(&c->ev[i]);
    for (int i = 0; i < 2 * CHIP_EVX; ++i) (void)hipEventCreate(&c->evx[i]);
    if (hipHostMalloc((void **)&c->h_pin, 64 * sizeof(u64), hipHostMallocDefault) != hipSuccess) {
        chip_set_error("hipHostMalloc failed");
        delete c;
        return CATCHHIP_ENOMEM;
    }
    *out = c;
    return 0;
}

void chip_phase_collect(catchhip_ctx *c, int phase) {
    float ms = 0.f;
    (void)hipEventSynchronize(c->ev[2 * phase + 1]);
    if (hipEventElapsedTime(&ms, c->ev[2 * phase], c->ev[2 * phase + 1]) == hipSuccess) c->phase_ms[phase] = ms;
}

int chip_pinned_reserve(catchhip_ctx *c, size_t bytes) {
    if (bytes <= c->h_big_bytes) return 0;
    size_t want = std::max<size_t>(bytes, (size_t)1 << 20);
    want = (want + 4095) & ~(size_t)4095;
    if (c->h_big) { (void)hipStreamSynchronize(c->stream); (void)hipHostFree(c->h_big); c->h_big = nullptr; c->h_big_bytes = 0; }
    if (hipHostMalloc(&c->h_big, want, hipHostMallocDefault) != hipSuccess) {
        chip_set_error("hipHostMalloc(%zu) failed", want);
        return CATCHHIP_ENOMEM;
    }
    c->h_big_bytes = want;
    return 0;
}

extern "C" int catchhip_ctx_destroy(catchhip_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)catchhip_comm_destroy(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    chip_pool_release_owner(c);
    for (int i = 0; i < 2 * NPHASE; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 2 * CHIP_EVX; ++i)
        if (c->evx[i]) (void)hipEventDestroy(c->evx[i]);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_big) (void)hipHostFree(c->h_big);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

extern "C" int catchhip_ctx_sync(catchhip_ctx *c) {
    ARG_CHECK(c != nullptr);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int catchhip_ctx_last_kernel_ms(catchhip_ctx *c, int phase, double *ms, i64 *launches) {
    ARG_CHECK(c != nullptr && phase >= 0 && phase < NPHASE);
    if ((phase == PHASE_VERIFY || phase == PHASE_VCOUNT) && c->phase_launches[phase]) chip_phase_collect(c, phase);   // recorded inside the scan phase
    if (ms) *ms = c->phase_ms[phase];
    if (launches) *launches = c->phase_launches[phase];
    return 0;
}

extern "C" int catchhip_ctx_last_counters(catchhip_ctx *c, i64 *out8) {
    ARG_CHECK(c != nullptr && out8 != nullptr);
    for (int i = 0; i < 8; ++i) out8[i] = c->counters[i];
    return 0;
}

extern "C" int catchhip_ctx_last_solver_counters(catchhip_ctx *c, i64 *out4) {
    ARG_CHECK(c != nullptr && out4 != nullptr);
    for (int i = 0; i < 4; ++i) out4[i] = c->solver_counters[i];
    return 0;
}

extern "C" int catchhip_ctx_last_ndf_counters(catchhip_ctx *c, i64 *out4) {
    ARG_CHECK(c != nullptr && out4 != nullptr);
    for (int i = 0; i < 4; ++i) out4[i] = c->ndf_counters[i];
    return 0;
}

extern "C" int catchhip_ctx_last_join_counters(catchhip_ctx *c, i64 *out4) {
    ARG_CHECK(c != nullptr && out4 != nullptr);
    for (int i = 0; i < 4; ++i) out4[i] = c->join_counters[i];
    return 0;
}

extern "C" int catchhip_ctx_last_seeds_dropped(catchhip_ctx *c, i64 *out) {
    ARG_CHECK(c != nullptr && out != nullptr);
    *out = c->seeds_dropped;
    return 0;
}

// ------------------------------------------------------------------------
// bit-plane packing: 32 bases per u32 word, 3 planes (code bit 0, 1, 2) with
// A=0 C=1 G=2 T=3 other(N)=4.  Base i of a stream lives in bit (i & 31) of
// word (i >> 5).  One wavefront packs 64 bases per step with three ballots.
// ------------------------------------------------------------------------
__device__ __forceinline__ u32 dna_code(u8 c) {
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}

// targets: SoA planes (plane b at planes + b*nwords)
__global__ void __launch_bounds__(256)
pack_targets_kernel(const u8 *__restrict__ bytes, i64 n, u32 *__restrict__ planes, i64 nwords,
                    uint4 *__restrict__ tq) {
    const int lane = threadIdx.x & 63;
    const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const i64 nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
    const i64 nchunks = (n + 63) >> 6;
    for (i64 ch = wave; ch < nchunks; ch += nwaves) {
        i64 i = ch * 64 + lane;
        u32 c = (i < n) ? dna_code(bytes[i]) : 0u;
        u64 b0 = __ballot(c & 1u), b1 = __ballot(c & 2u), b2 = __ballot(c & 4u);
        if (lane < 6) {
            u64 b = lane < 2 ? b0 : (lane < 4 ? b1 : b2);
            u32 w = (lane & 1) ? (u32)(b >> 32) : (u32)b;
            i64 wi = ch * 2 + (lane & 1);
            if (wi < nwords) planes[(size_t)(lane >> 1) * nwords + wi] = w;
        }
        if (lane < 2) {   // word-interleaved copy
            const i64 wi = ch * 2 + lane;
            if (wi < nwords)
                tq[wi] = lane ? make_uint4((u32)(b0 >> 32), (u32)(b1 >> 32), (u32)(b2 >> 32), 0u)
                              : make_uint4((u32)b0, (u32)b1, (u32)b2, 0u);
        }
    }
}

// probes: [probe][word][4] (x = plane0, y = plane1, z = plane2, w = 0), one
// wavefront per probe (probes are short: L <= 256)
__global__ void __launch_bounds__(256)
pack_probes_kernel(const u8 *__restrict__ bytes, const u32 *__restrict__ probe_off, i64 nprobes,
                   int pwords, u32 *__restrict__ planes) {
    const int lane = threadIdx.x & 63;
    const i64 p = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (p >= nprobes) return;
    const u32 o = probe_off[p];
    const int L = (int)(probe_off[p + 1] - o);
    for (int ch = 0; ch * 64 < L; ++ch) {
        int i = ch * 64 + lane;
        u32 c = (i < L) ? dna_code(bytes[o + i]) : 0u;
        u64 b0 = __ballot(c & 1u), b1 = __ballot(c & 2u), b2 = __ballot(c & 4u);
        if (lane < 8) {
            int half = lane >> 2, comp = lane & 3;
            int wi = ch * 2 + half;
            u64 b = comp == 0 ? b0 : (comp == 1 ? b1 : (comp == 2 ? b2 : 0ull));
            u32 w = half ? (u32)(b >> 32) : (u32)b;
            if (wi < pwords) planes[((size_t)p * pwords + wi) * 4 + comp] = w;
        }
    }
}

// word 0 of planes 0/1 per probe, contiguous: the hot loop of the scan reads
// it through the scalar cache (one s_load per probe, operands in SGPRs)
__global__ void __launch_bounds__(256)
probe_w0_kernel(const u32 *__restrict__ planes, i64 nprobes, int pwords, u32 mask0, uint2 *__restrict__ w0) {
    i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprobes) return;
    const u32 *q = planes + (size_t)p * pwords * 4;
    w0[p] = make_uint2(q[0] & mask0, q[1] & mask0);
}

static void alphabet_scan(const u8 *b, i64 n, bool *dna5, bool *has_n) {
    bool present[256] = {false};
    for (i64 i = 0; i < n; ++i) present[b[i]] = true;
    *dna5 = true;
    *has_n = false;
    for (int c = 0; c < 256; ++c) {
        if (!present[c]) continue;
        if (c == 'A' || c == 'C' || c == 'G' || c == 'T') continue;
        *has_n = true;
        if (c != 'N') *dna5 = false;
    }
}

// alpha: null, or {dna5, has_n} already known for `bytes` (catchhip_targets_create_ptrs)
static int targets_create_impl(catchhip_ctx *ctx, const u8 *bytes, const i64 *seq_off, const i32 *seq_genome,
                               i64 nseq, i32 ngenomes, const bool *alpha, catchhip_targets **out) {
    ARG_CHECK(ctx && out && seq_off && nseq >= 0 && ngenomes >= 0);
    PoolScope pool_scope(ctx);
    ARG_CHECK(nseq == 0 || (bytes != nullptr && seq_genome != nullptr) || seq_off[nseq] == 0);
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    i64 total = seq_off[nseq];
    ARG_CHECK(seq_off[0] == 0 && total >= 0);
    if (total >= ((i64)1 << 32) - 4096) {
        chip_set_error("targets larger than 2^32 bases per group are not supported");
        return CATCHHIP_EINVAL;
    }
    catchhip_targets *t = new catchhip_targets();
    t->ctx = ctx;
    t->total = total;
    t->nseq = nseq;
    t->ngenomes = ngenomes;
    t->h_seq_off.assign(seq_off, seq_off + nseq + 1);
    t->h_seq_genome.assign(seq_genome, seq_genome + nseq);
    t->min_seq_len = nseq ? ((i64)1 << 62) : 0;
    std::vector<u32> so32((size_t)nseq + 1);
    for (i64 i = 0; i <= nseq; ++i) so32[i] = (u32)seq_off[i];
    t->h_genome_off.assign((size_t)ngenomes + 1, -1);
    i32 prev = -1;
    for (i64 i = 0; i < nseq; ++i) {
        i64 len = seq_off[i + 1] - seq_off[i];
        i32 g = seq_genome[i];
        if (len < 0 || g < prev || g < 0 || g >= ngenomes) {
            delete t;
            chip_set_error("targets: bad seq_off / seq_genome (must be non-decreasing)");
            return CATCHHIP_EINVAL;
        }
        if (len < t->min_seq_len) t->min_seq_len = len;
        if (g != prev) t->h_genome_off[g] = seq_off[i];
        prev = g;
    }
    // genomes without sequences get an empty range
    t->h_genome_off[ngenomes] = total;
    for (i32 g = ngenomes - 1; g >= 0; --g)
        if (t->h_genome_off[g] < 0) t->h_genome_off[g] = t->h_genome_off[g + 1];
    std::vector<u32> go32((size_t)ngenomes + 1);
    for (i32 g = 0; g <= ngenomes; ++g) go32[g] = (u32)t->h_genome_off[g];
    if (alpha) { t->dna5 = alpha[0]; t->has_n = alpha[1]; }
    else alphabet_scan(bytes, total, &t->dna5, &t->has_n);

    int rc = 0;
    do {
        if ((rc = t->bytes.alloc((size_t)total + 256))) break;
        if ((rc = t->seq_off.alloc((size_t)nseq + 1))) break;
        if ((rc = t->seq_genome.alloc((size_t)nseq))) break;
        if ((rc = t->genome_off.alloc((size_t)ngenomes + 1))) break;
        hipStream_t s = ctx->stream;
        if (hipMemsetAsync(t->bytes.p, 0, (size_t)total + 256, s) != hipSuccess ||
            (total && hipMemcpyAsync(t->bytes.p, bytes, (size_t)total, hipMemcpyHostToDevice, s) != hipSuccess) ||
            hipMemcpyAsync(t->seq_off.p, so32.data(), sizeof(u32) * (nseq + 1), hipMemcpyHostToDevice, s) != hipSuccess ||
            (nseq && hipMemcpyAsync(t->seq_genome.p, seq_genome, sizeof(i32) * nseq, hipMemcpyHostToDevice, s) != hipSuccess) ||
            hipMemcpyAsync(t->genome_off.p, go32.data(), sizeof(u32) * (ngenomes + 1), hipMemcpyHostToDevice, s) != hipSuccess) {
            chip_set_error("targets upload failed");
            rc = CATCHHIP_EHIP;
            break;
        }
        if (t->dna5) {
            // room for a whole scan tile of overhang beyond the last base
            t->nwords = (total + 8192) / 32 + 64;
            if ((rc = t->planes.alloc((size_t)t->nwords * 3))) break;
            if (hipMemsetAsync(t->planes.p, 0, sizeof(u32) * t->nwords * 3, s) != hipSuccess) { rc = CATCHHIP_EHIP; break; }
            if ((rc = t->tq.alloc((size_t)t->nwords))) break;
            if (hipMemsetAsync(t->tq.p, 0, sizeof(uint4) * t->nwords, s) != hipSuccess) { rc = CATCHHIP_EHIP; break; }
            if (total) {
                i64 chunks = (total + 63) / 64;
                unsigned blocks = (unsigned)(chunks < 4 * 2048 ? div_up(chunks, 4) : 2048);
                hipLaunchKernelGGL(pack_targets_kernel, dim3(blocks), dim3(256), 0, s, t->bytes.p, total,
                                   t->planes.p, t->nwords, t->tq.p);
            }
        }
        if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) {
            chip_set_error("targets pack failed");
            rc = CATCHHIP_EHIP;
            break;
        }
    } while (0);
    if (rc) { delete t; return rc; }
    *out = t;
    return 0;
}

extern "C" int catchhip_targets_create(catchhip_ctx *ctx, const u8 *bytes, const i64 *seq_off,
                                       const i32 *seq_genome, i64 nseq, i32 ngenomes,
                                       catchhip_targets **out) {
    return targets_create_impl(ctx, bytes, seq_off, seq_genome, nseq, ngenomes, nullptr, out);
}

// The same from one pointer per sequence (the host holds its sequences as
// separate strings: joining and re-encoding 592 MB of them in Python was 0.22 s
// of the 0.31 s a bench upload took).  The sequences are gathered into pinned
// memory by a few host threads, which also look at the alphabet, and uploaded
// from there.
#include <chrono>
#include <thread>
extern "C" int catchhip_targets_create_ptrs(catchhip_ctx *ctx, const u8 *const *seq_ptr, const i64 *seq_len,
                                            const i32 *seq_genome, i64 nseq, i32 ngenomes,
                                            catchhip_targets **out) {
    ARG_CHECK(ctx && out && nseq >= 0 && ngenomes >= 0 && (nseq == 0 || (seq_ptr && seq_len && seq_genome)));
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<i64> off((size_t)nseq + 1, 0);
    for (i64 i = 0; i < nseq; ++i) {
        ARG_CHECK(seq_len[i] >= 0 && (seq_len[i] == 0 || seq_ptr[i] != nullptr));
        off[i + 1] = off[i] + seq_len[i];
    }
    const i64 total = off[nseq];
    if (total >= ((i64)1 << 32) - 4096) {
        chip_set_error("targets larger than 2^32 bases per group are not supported");
        return CATCHHIP_EINVAL;
    }
    TRY(chip_pinned_reserve(ctx, (size_t)total + 64));
    u8 *stage = (u8 *)ctx->h_big;
    // one thread per 2 MB, at most 16 (the copy runs at memory speed from a handful of cores; the
    // alphabet test is three compares per byte that the compiler vectorises)
    static const int max_threads = getenv("CATCHHIP_GATHER_THREADS") ? std::max(1, atoi(getenv("CATCHHIP_GATHER_THREADS"))) : 16;
    const int nthreads = (int)std::max<i64>(1, std::min<i64>(max_threads, total >> 21));
    std::vector<unsigned char> seen((size_t)nthreads * 2, 0);   // per thread: {an N, some other symbol}
    const auto t_g0 = std::chrono::steady_clock::now();
    auto work = [&](int tix) {
        // thread tix takes the sequences whose bytes start in its slice of the total
        const i64 lo = total * tix / nthreads, hi = total * (tix + 1) / nthreads;
        i64 i = std::upper_bound(off.begin(), off.end(), lo) - off.begin() - 1;
        if (i < 0) i = 0;
        unsigned char any_n = 0, any_other = 0;
        for (; i < nseq && off[i] < hi; ++i) {
            if (off[i] < lo) continue;          // belongs to the previous slice
            const u8 *src = seq_ptr[i];
            const i64 n = seq_len[i];
            u8 *dst = stage + off[i];
            memcpy(dst, src, (size_t)n);
            unsigned char nn = 0, oo = 0;
            for (i64 j = 0; j < n; ++j) {
                const u8 c = dst[j];
                const unsigned char acgt = (unsigned char)((c == 'A') | (c == 'C') | (c == 'G') | (c == 'T'));
                const unsigned char isn = (unsigned char)(c == 'N');
                nn |= isn;
                oo |= (unsigned char)!(acgt | isn);
            }
            any_n |= nn; any_other |= oo;
        }
        seen[(size_t)tix * 2] = any_n; seen[(size_t)tix * 2 + 1] = any_other;
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int tix = 0; tix < nthreads; ++tix) th.emplace_back(work, tix);
        for (auto &x : th) x.join();
    }
    bool alpha[2] = {true, false};   // {subset of ACGTN, something other than ACGT}
    for (int tix = 0; tix < nthreads; ++tix) {
        if (seen[(size_t)tix * 2] || seen[(size_t)tix * 2 + 1]) alpha[1] = true;
        if (seen[(size_t)tix * 2 + 1]) alpha[0] = false;
    }
    if (getenv("CATCHHIP_TIMING"))
        fprintf(stderr, "[catchhip] targets: gathered %lld bytes on %d threads in %.3f ms\n", (long long)total, nthreads,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_g0).count());
    const int rc = targets_create_impl(ctx, stage, off.data(), seq_genome, nseq, ngenomes, alpha, out);
    return rc;
}

// Hand-over between contexts of one device (an upload context packs the next
// group while the compute context's stream works on the current one): waits
// for the stream the object was built on, after which it belongs to `to`.  Its
// device blocks keep their pool owner and return to the builder's cache when
// the object is destroyed -- by then every user of them has finished.
static int rebind_check(catchhip_ctx *from, catchhip_ctx *to) {
    ARG_CHECK(from && to && from->device == to->device);
    HIP_TRY(hipSetDevice(from->device));
    if (from != to) HIP_TRY(hipStreamSynchronize(from->stream));
    return 0;
}
extern "C" int catchhip_targets_rebind(catchhip_targets *t, catchhip_ctx *to) {
    ARG_CHECK(t != nullptr);
    TRY(rebind_check(t->ctx, to));
    t->ctx = to;
    return 0;
}
extern "C" int catchhip_probes_rebind(catchhip_probes *p, catchhip_ctx *to) {
    ARG_CHECK(p != nullptr);
    TRY(rebind_check(p->ctx, to));
    p->ctx = to;
    return 0;
}

extern "C" int catchhip_targets_destroy(catchhip_targets *t) {
    if (t) { (void)hipSetDevice(t->ctx->device); delete t; }
    return 0;
}

// packed image of equal-length DNA probes (planes + word 0), from p->bytes / p->probe_off
int chip_probes_pack_planes(catchhip_probes *p) {
    hipStream_t s = p->ctx->stream;
    const i64 nprobes = p->nprobes;
    if (!(p->dna5 && p->L > 0 && p->L <= 256)) return 0;
    p->pwords = (p->L + 31) / 32;
    size_t nw = (size_t)nprobes * p->pwords * 4;
    TRY(p->planes.alloc(nw + 1024));
    HIP_TRY(hipMemsetAsync(p->planes.p, 0, sizeof(u32) * (nw + 1024), s));
    unsigned blocks = (unsigned)div_up(nprobes, 4);
    if (nprobes)
        hipLaunchKernelGGL(pack_probes_kernel, dim3(blocks), dim3(256), 0, s, p->bytes.p, p->probe_off.p, nprobes,
                           p->pwords, p->planes.p);
    TRY(p->w0.alloc((size_t)nprobes + 64));
    HIP_TRY(hipMemsetAsync(p->w0.p, 0, sizeof(uint2) * (nprobes + 64), s));
    const u32 mask0 = (p->pwords == 1 && (p->L & 31)) ? ((1u << (p->L & 31)) - 1u) : 0xffffffffu;
    if (nprobes)
        hipLaunchKernelGGL(probe_w0_kernel, dim3((unsigned)div_up(nprobes, 256)), dim3(256), 0, s, p->planes.p,
                           nprobes, p->pwords, mask0, p->w0.p);
    return 0;
}

extern "C" int catchhip_probes_create(catchhip_ctx *ctx, const u8 *bytes, const i64 *probe_off,
                                      i64 nprobes, const i32 *set_id, const i32 *ent_probe,
                                      const i32 *ent_pos, i64 nent, i32 k, catchhip_probes **out) {
    ARG_CHECK(ctx && out && probe_off && nprobes >= 0 && nent >= 0);
    PoolScope pool_scope(ctx);
    ARG_CHECK(nprobes == 0 || (bytes && set_id));
    ARG_CHECK(nent == 0 || (ent_probe && ent_pos && k > 0));
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    i64 total = probe_off[nprobes];
    ARG_CHECK(probe_off[0] == 0 && total >= 0 && total < ((i64)1 << 31));
    catchhip_probes *p = new catchhip_probes();
    p->ctx = ctx;
    p->nprobes = nprobes;
    p->total = total;
    p->nent = nent;
    p->k = k;
    p->L = nprobes ? (i32)(probe_off[1] - probe_off[0]) : 0;
    std::vector<u32> po32((size_t)nprobes + 1);
    for (i64 i = 0; i <= nprobes; ++i) po32[i] = (u32)probe_off[i];
    for (i64 i = 0; i < nprobes; ++i) {
        i64 len = probe_off[i + 1] - probe_off[i];
        if (set_id[i] < 0) { delete p; chip_set_error("probes: negative set id"); return CATCHHIP_EINVAL; }
        if (set_id[i] > p->max_set_id) p->max_set_id = set_id[i];
        if (len <= 0) { delete p; chip_set_error("probes: empty probe"); return CATCHHIP_EINVAL; }
        if (len != p->L) p->L = -1;
    }
    // anchors must lie inside their probe
    std::vector<i32> per_probe((size_t)nprobes, 0);
    bool pigeon = (p->L > 0 && k > 0 && p->L % k == 0);
    for (i64 e = 0; e < nent; ++e) {
        i32 q = ent_probe[e], a = ent_pos[e];
        if (q < 0 || q >= nprobes || a < 0 || a + k > probe_off[q + 1] - probe_off[q]) {
            delete p;
            chip_set_error("probes: anchor %lld out of range", (long long)e);
            return CATCHHIP_EINVAL;
        }
        if (pigeon) {
            if (a % k != 0) pigeon = false;
            else per_probe[q]++;
        }
    }
    if (pigeon) {
        // unique entries + every multiple of k present <=> count == L/k each
        i32 want = p->L / k;
        for (i64 i = 0; i < nprobes && pigeon; ++i)
            if (per_probe[i] != want) pigeon = false;
        if (nent != (i64)want * nprobes) pigeon = false;
    }
    p->pigeonhole = pigeon && nprobes > 0;
    p->sorted_unique = true;
    for (i64 e = 1; e < nent && p->sorted_unique; ++e)
        if (ent_probe[e] < ent_probe[e - 1] || (ent_probe[e] == ent_probe[e - 1] && ent_pos[e] <= ent_pos[e - 1]))
            p->sorted_unique = false;
    alphabet_scan(bytes, total, &p->dna5, &p->has_n);

    int rc = 0;
    do {
        hipStream_t s = ctx->stream;
        if ((rc = p->bytes.alloc((size_t)total + 256))) break;
        if ((rc = p->probe_off.alloc((size_t)nprobes + 1))) break;
        if ((rc = p->set_id.alloc((size_t)nprobes))) break;
        if ((rc = p->ent_probe.alloc((size_t)nent))) break;
        if ((rc = p->ent_pos.alloc((size_t)nent))) break;
        if (hipMemsetAsync(p->bytes.p, 0, (size_t)total + 256, s) != hipSuccess ||
            (total && hipMemcpyAsync(p->bytes.p, bytes, (size_t)total, hipMemcpyHostToDevice, s) != hipSuccess) ||
            hipMemcpyAsync(p->probe_off.p, po32.data(), sizeof(u32) * (nprobes + 1), hipMemcpyHostToDevice, s) != hipSuccess ||
            (nprobes && hipMemcpyAsync(p->set_id.p, set_id, sizeof(i32) * nprobes, hipMemcpyHostToDevice, s) != hipSuccess) ||
            (nent && hipMemcpyAsync(p->ent_probe.p, ent_probe, sizeof(i32) * nent, hipMemcpyHostToDevice, s) != hipSuccess) ||
            (nent && hipMemcpyAsync(p->ent_pos.p, ent_pos, sizeof(i32) * nent, hipMemcpyHostToDevice, s) != hipSuccess)) {
            chip_set_error("probes upload failed");
            rc = CATCHHIP_EHIP;
            break;
        }
        {   // buckets of the row build: unique probes grouped by set id, in set-id order
            std::vector<i32> ids(set_id, set_id + nprobes);
            std::sort(ids.begin(), ids.end());
            ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
            p->nbuckets = (i64)ids.size();
            std::vector<u32> bo((size_t)nprobes);
            for (i64 i = 0; i < nprobes; ++i)
                bo[i] = (u32)(std::lower_bound(ids.begin(), ids.end(), set_id[i]) - ids.begin());
            if ((rc = p->bucket_of.alloc((size_t)nprobes + 1))) break;
            if ((rc = p->bucket_set.alloc(ids.size() + 1))) break;
            if (nprobes && (hipMemcpyAsync(p->bucket_of.p, bo.data(), sizeof(u32) * nprobes, hipMemcpyHostToDevice, s) != hipSuccess ||
                            hipMemcpyAsync(p->bucket_set.p, ids.data(), sizeof(i32) * ids.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
                            hipStreamSynchronize(s) != hipSuccess)) {
                chip_set_error("probes upload failed");
                rc = CATCHHIP_EHIP;
                break;
            }
        }
        if (nent) {   // anchors grouped by probe, ascending position
            std::vector<u64> key((size_t)nent);
            for (i64 e = 0; e < nent; ++e) key[e] = ((u64)(u32)ent_probe[e] << 32) | (u32)ent_pos[e];
            std::sort(key.begin(), key.end());
            std::vector<u32> sp((size_t)nent), so((size_t)nent), ptr((size_t)nprobes + 1, 0);
            for (i64 e = 0; e < nent; ++e) { sp[e] = (u32)(key[e] >> 32); so[e] = (u32)key[e]; ptr[sp[e] + 1]++; }
            for (i64 i = 0; i < nprobes; ++i) ptr[i + 1] += ptr[i];
            if ((rc = p->sent_probe.alloc((size_t)nent))) break;
            if ((rc = p->sent_pos.alloc((size_t)nent))) break;
            if ((rc = p->ent_ptr.alloc((size_t)nprobes + 1))) break;
            if (hipMemcpyAsync(p->sent_probe.p, sp.data(), sizeof(u32) * nent, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipMemcpyAsync(p->sent_pos.p, so.data(), sizeof(u32) * nent, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipMemcpyAsync(p->ent_ptr.p, ptr.data(), sizeof(u32) * (nprobes + 1), hipMemcpyHostToDevice, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess) {
                chip_set_error("probes upload failed");
                rc = CATCHHIP_EHIP;
                break;
            }
        }
        if ((rc = chip_probes_pack_planes(p))) break;
        if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) {
            chip_set_error("probes pack failed");
            rc = CATCHHIP_EHIP;
            break;
        }
    } while (0);
    if (rc) { delete p; return rc; }
    *out = p;
    return 0;
}

extern "C" int catchhip_probes_destroy(catchhip_probes *p) {
    if (p) { (void)hipSetDevice(p->ctx->device); delete p; }
    return 0;
}

extern "C" int catchhip_rows_destroy(catchhip_rows *r) {
    if (r) { (void)hipSetDevice(r->ctx->device); delete r; }
    return 0;
}

// ---- iteration order of a CPython set ---------------------------------------
// The reference's near-duplicate filters return `list(to_include)`, a SET of probes
// (catch/filter/near_duplicate_filter.py:76-103), and the set cover filter numbers its
// candidates in the order it is handed them: which of two equally good probes is picked
// follows from the set's iteration order.  That order is a function of the keys' hashes
// and of the order they were added in (Objects/setobject.c, CPython 3.7-3.12: open
// addressing, home slot hash & mask, 9 linear probes, then i = i * 5 + 1 + (perturb >>= 5);
// a table of 8 slots rebuilt -- entries re-inserted in slot order -- when fill * 5 >=
// mask * 3, to the smallest power of two above 4 x used, 2 x used beyond 50,000), and
// list(set) walks the table by slot.  Only insertions of distinct keys occur here.
void chip_pyset_order(const i64 *hash, i64 n, i64 *order) {
    size_t size = 8, fill = 0;
    std::vector<i64> table(size, -1);
    auto insert_clean = [&](std::vector<i64> &tab, size_t mask, i64 idx) {
        const u64 h = (u64)hash[idx];
        size_t perturb = (size_t)h, i = (size_t)h & mask;
        for (;;) {
            if (tab[i] < 0) { tab[i] = idx; return; }
            if (i + 9 <= mask)
                for (size_t j = 1; j <= 9; ++j)
                    if (tab[i + j] < 0) { tab[i + j] = idx; return; }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    };
    for (i64 idx = 0; idx < n; ++idx) {
        const size_t mask = size - 1;
        insert_clean(table, mask, idx);
        ++fill;
        if (fill * 5 >= mask * 3) {
            const size_t want = fill > 50000 ? fill * 2 : fill * 4;
            size_t nsize = 8;
            while (nsize <= want) nsize <<= 1;
            std::vector<i64> nt(nsize, -1);
            for (size_t sl = 0; sl < size; ++sl)
                if (table[sl] >= 0) insert_clean(nt, nsize - 1, table[sl]);
            table.swap(nt);
            size = nsize;
        }
    }
    i64 at = 0;
    for (size_t sl = 0; sl < size; ++sl)
        if (table[sl] >= 0) order[at++] = table[sl];
}

extern "C" int catchhip_pyset_order(const i64 *hashes, i64 n, i64 *order) {
    ARG_CHECK(n >= 0 && (n == 0 || (hashes && order)));
    chip_pyset_order(hashes, n, order);
    return 0;
}

extern "C" int catchhip_pyset_order_strs(const u8 *bytes, const i64 *off, i64 n, i64 *order) {
    ARG_CHECK(n >= 0 && (n == 0 || (bytes && off && order)));
    std::vector<i64> h((size_t)n);
    for (i64 i = 0; i < n; ++i) {
        ARG_CHECK(off[i + 1] >= off[i] && off[i + 1] - off[i] < ((i64)1 << 31));
        h[(size_t)i] = chip_pyhash_seed0(bytes + off[i], (int)(off[i + 1] - off[i]));
    }
    chip_pyset_order(h.data(), n, order);
    return 0;
}

// ---- independent instances inside one probes / targets pair ----------------
extern "C" int catchhip_probes_set_groups(catchhip_ctx *ctx, catchhip_probes *P, const i32 *group_of_probe) {
    ARG_CHECK(ctx && P && P->ctx == ctx);
    PoolScope pool_scope(ctx);
    if (!group_of_probe) { P->has_groups = false; return 0; }
    HIP_TRY(hipSetDevice(ctx->device));
    TRY(P->group.alloc((size_t)P->nprobes + 1));
    if (P->nprobes) {
        HIP_TRY(hipMemcpyAsync(P->group.p, group_of_probe, sizeof(i32) * (size_t)P->nprobes, hipMemcpyHostToDevice,
                               ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    P->has_groups = true;
    return 0;
}

extern "C" int catchhip_targets_set_groups(catchhip_ctx *ctx, catchhip_targets *T, const i32 *group_of_genome) {
    ARG_CHECK(ctx && T && T->ctx == ctx);
    PoolScope pool_scope(ctx);
    if (!group_of_genome) { T->has_groups = false; return 0; }
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<i32> sg((size_t)T->nseq + 1, 0);
    T->ngroups_set = 0;
    for (i32 g = 0; g < T->ngenomes; ++g) {
        ARG_CHECK(group_of_genome[g] >= 0);
        T->ngroups_set = std::max(T->ngroups_set, group_of_genome[g] + 1);
    }
    for (i64 s = 0; s < T->nseq; ++s) sg[(size_t)s] = group_of_genome[T->h_seq_genome[(size_t)s]];
    TRY(T->seq_group.alloc((size_t)T->nseq + 1));
    HIP_TRY(hipMemcpyAsync(T->seq_group.p, sg.data(), sizeof(i32) * ((size_t)T->nseq + 1), hipMemcpyHostToDevice,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    T->has_groups = true;
    return 0;
}
