// Fused entry points: cover scan + greedy solve per group, and several
// independent groups at once on their own streams
// (catch/filter/set_cover_filter.py:816-846 per group).
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"

extern "C" int catchhip_setcover_filter(catchhip_ctx *ctx, const catchhip_probes *P, const catchhip_targets *T,
                                        i32 mismatches, i32 lcf_thres, i32 island, i32 cover_extension, i32 mode,
                                        i64 num_sets, const i64 *ranks, const double *universe_p, i64 *out_ids,
                                        i64 *n_out, i64 *nrows) {
    ARG_CHECK(ctx && P && T && n_out);
    ARG_CHECK(P->ctx == ctx && T->ctx == ctx && cover_extension >= 0 && num_sets >= 0);
    catchhip_rows *R = nullptr;
    i64 nr = 0;
    int rc;
    // Full coverage of every universe (the default -c 1.0) on the seed path:
    // scan, row build and the first batch of solver rounds are queued without a
    // single host synchronisation; the solver checks on the device that the scan
    // did not overflow and the rows fit its 5-word path, otherwise the group is
    // redone through the two synchronous calls below.
    bool full = ctx->comm == nullptr && !chip_test_env("CATCHHIP_GREEDY_SEQUENTIAL") && num_sets > 0 && out_ids;
    if (universe_p)
        for (i32 u = 0; u < T->ngenomes && full; ++u) full = universe_p[u] == 1.0;
    const auto t_in = std::chrono::steady_clock::now();
    if (full) {
        rc = chip_cover_scan_nosync(ctx, P, T, mismatches, lcf_thres, island, cover_extension, mode, &R);
        if (rc < 0) return rc;
        if (rc == 0) {
            int retry = 0;
            rc = chip_greedy_deferred(ctx, R, num_sets, ranks, out_ids, n_out, &retry);
            if (R->seed_ratio_seen > P->seed_ratio_hint) P->seed_ratio_hint = R->seed_ratio_seen;   // sizes the next work list
            (void)catchhip_rows_destroy(R);
            R = nullptr;
            if (rc) return rc;
            if (!retry) {
                if (nrows) *nrows = ctx->counters[7];
                if (getenv("CATCHHIP_TIMING"))
                    fprintf(stderr, "[catchhip] fused filter: %.1f us in the call\n",
                            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_in).count());
                return 0;
            }
        }
    }
    const bool timing = getenv("CATCHHIP_TIMING") != nullptr;   // host wall time of the two halves (stderr)
    const auto t0 = std::chrono::steady_clock::now();
    rc = catchhip_cover_scan(ctx, P, T, mismatches, lcf_thres, island, cover_extension, mode, &R, &nr);
    if (rc) return rc;
    if (nrows) *nrows = nr;
    const auto t1 = std::chrono::steady_clock::now();
    rc = catchhip_setcover_greedy(ctx, R, num_sets, ranks, universe_p, out_ids, n_out);
    const auto t2 = std::chrono::steady_clock::now();
    (void)catchhip_rows_destroy(R);
    if (timing)
        fprintf(stderr, "[catchhip] filter: scan %.3f ms, solve %.3f ms (host wall)\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t2 - t1).count());
    return rc;
}

// Persistent helper threads for catchhip_setcover_filter_many: creating and
// joining a std::thread per group cost ~50 us per call, a tenth of an S2 step.
// Helper h sleeps on a condition variable until a job generation is posted.
namespace {
struct ManyPool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> helpers;
    std::function<void(int)> job;   // argument: group index
    int first = 0, count = 0;       // groups [first, first+count) belong to the helpers
    int next = 0, done = 0;
    unsigned long long generation = 0;
    bool stop = false;
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            int g;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || (generation != seen && next < count) || generation != seen; });
                if (stop) return;
                if (next >= count) { seen = generation; continue; }
                g = first + next++;
            }
            job(g);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (++done == count) cv_done.notify_all();
            }
        }
    }
    void run(int first_group, int ngroups, const std::function<void(int)> &fn) {
        std::unique_lock<std::mutex> lk(mu);
        while ((int)helpers.size() < ngroups) helpers.emplace_back([this] { loop(); });
        job = fn; first = first_group; count = ngroups; next = 0; done = 0;
        ++generation;
        lk.unlock();
        cv_work.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return done == count; });
    }
    ~ManyPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_work.notify_all();
        for (auto &t : helpers) t.join();
    }
};
ManyPool g_many_pool;
std::mutex g_many_call;   // one _many call at a time uses the pool
}  // namespace

extern "C" int catchhip_setcover_filter_many(i32 n, catchhip_ctx *const *ctxs, const catchhip_probes *const *probes,
                                             const catchhip_targets *const *targets, i32 mismatches, i32 lcf_thres,
                                             i32 island, i32 cover_extension, i32 mode, const i64 *num_sets,
                                             const i64 *const *ranks, const double *const *universe_p,
                                             i64 *const *out_ids, i64 *n_out, i64 *nrows) {
    ARG_CHECK(n >= 0 && (n == 0 || (ctxs && probes && targets && num_sets && out_ids && n_out)));
    for (i32 g = 0; g < n; ++g)
        for (i32 h = 0; h < g; ++h)
            if (ctxs[g] == ctxs[h]) { chip_set_error("setcover_filter_many: contexts must be distinct"); return CATCHHIP_EINVAL; }
    std::vector<int> rcs((size_t)n, 0);
    std::vector<std::string> msgs((size_t)n);
    auto work = [&](i32 g) {
        rcs[g] = catchhip_setcover_filter(ctxs[g], probes[g], targets[g], mismatches, lcf_thres, island,
                                          cover_extension, mode, num_sets[g], ranks ? ranks[g] : nullptr,
                                          universe_p ? universe_p[g] : nullptr, out_ids[g], &n_out[g],
                                          nrows ? &nrows[g] : nullptr);
        if (rcs[g]) msgs[g] = catchhip_last_error();   // thread-local: carry it to the caller
    };
    if (n > 1) {
        std::lock_guard<std::mutex> call(g_many_call);
        g_many_pool.run(1, n - 1, work);
        work(0);
        g_many_pool.wait();
    } else if (n == 1) {
        work(0);
    }
    for (i32 g = 0; g < n; ++g)
        if (rcs[g]) { chip_set_error("%s", msgs[g].c_str()); return rcs[g]; }
    return 0;
}
