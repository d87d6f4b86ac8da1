/*
 * catchhip.h -- C ABI of libcatchhip.so: the MI355X (gfx950) implementation of
 * the probe-coverage / set-cover / near-duplicate hot path of
 * broadinstitute/catch v1.5.2.
 *
 * The reference is pure Python and has no FFI of its own: the functions below
 * are what a ctypes binding for this path binds (INTEGRATION.md shows the
 * stub).  Each entry point cites the reference code (relative to the
 * reference repository root) whose behaviour it replaces.  Plain pointers and
 * sizes only; every buffer named `const T* x` is a caller-owned HOST buffer
 * unless the comment says "device".  All functions return 0 on success and a
 * negative CATCHHIP_E* code on failure; catchhip_last_error() gives the
 * message for the calling thread.  One host thread per context.
 *
 * Coordinates.  A `targets` object is a list of sequences, concatenated in
 * the order given; consecutive sequences with the same genome index form one
 * genome (= one "universe" of the set cover, catch/filter/
 * set_cover_filter.py:414-453).  Cover rows are reported in genome
 * coordinates (position within the genome's concatenated sequences), exactly
 * as SetCoverFilter._make_sets builds them.
 */
#ifndef CATCHHIP_H
#define CATCHHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CATCHHIP_ABI_VERSION 1

#define CATCHHIP_OK 0
#define CATCHHIP_EINVAL (-1)   /* bad argument */
#define CATCHHIP_EHIP (-2)     /* HIP runtime error */
#define CATCHHIP_ENOMEM (-3)   /* allocation failed */
#define CATCHHIP_ERANK (-4)    /* rank list exhausted (reference: IndexError) */
#define CATCHHIP_ECOMM (-5)    /* RCCL error */

typedef struct catchhip_ctx catchhip_ctx;
typedef struct catchhip_targets catchhip_targets;
typedef struct catchhip_probes catchhip_probes;
typedef struct catchhip_rows catchhip_rows;

int catchhip_abi_version(void);
const char *catchhip_last_error(void);

/* ---- context: one HIP device + one stream ----------------------------- */
int catchhip_device_count(int *count);
int catchhip_ctx_create(int device, catchhip_ctx **out);
int catchhip_ctx_destroy(catchhip_ctx *ctx);
int catchhip_ctx_sync(catchhip_ctx *ctx);
/* Device-memory cache of the library (all contexts): out4 = {hipMalloc calls so
 * far, bytes held from the driver, bytes of those currently free in the cache,
 * hipMalloc failures answered by emptying the caller's cache}.  No reference
 * counterpart (the reference keeps its working set in Python objects); the bench
 * reports it so that a step that allocates is visible. */
int catchhip_pool_stats(int64_t *out4);
/* Returns every idle cached block of every context of the process to the driver, after a synchronisation of
 * the current device (blocks of other devices: hipFree itself waits for the device that owns the block).  For the seams of a long-running process whose next
 * phase allocates differently (the cache is keyed by context and size class). */
int catchhip_pool_trim(void);
/* Elapsed GPU milliseconds spent in the kernels of the most recent call of
 * the named phase (HIP events on the context's stream).  phase: 0 = cover
 * scan kernels (K1 hit search), 1 = row build (sort/merge), 2 = greedy
 * set-cover kernels (set-up + solver), 3 = near-duplicate kernels, 4 = only the
 * (count+claim, check+apply) launch pairs of the frontier solver's rounds,
 * 5 = only the seed-verify launch of the seed scan (part of phase 0),
 * 6 = only the claim launches of the row-parallel solver (part of phase 4;
 * 0 launches when the last solve used the other kernels).
 * *launches = kernel launches timed. */
int catchhip_ctx_last_kernel_ms(catchhip_ctx *ctx, int phase, double *ms,
                                int64_t *launches);
/* Work counters of the most recent calls (for roofline accounting), 8 values:
 * [0] raw hits found by the last cover scan, [1] seeds verified (seed scan /
 * seed join), [2] greedy iterations (rounds), [3] picks, [4] winner rows applied
 * (sequential solver; row-parallel solver: row records streamed), [5] rows
 * re-counted by the greedy solver, [6] bitmap
 * words read while re-counting, [7] cover rows of the last fused
 * catchhip_setcover_filter call. */
int catchhip_ctx_last_counters(catchhip_ctx *ctx, int64_t *out8);
/* Work of the last key-grouped join (the seed scan of pigeonhole tables, csrc/scan_join.inc): out4[0] = target
 * positions whose k-mer is in the anchor table, [1] = (position, entry) pairs verified (the table matches of
 * catch/probe.py:1062-1069), [2] = lane slots the wave-wide runs occupied (64 per window chunk and entry; pairs /
 * slots = lane utilisation), [3] = tasks of runs that were cut by entries.  Zeros after a seed-list scan. */
int catchhip_ctx_last_join_counters(catchhip_ctx *ctx, int64_t *out4);

/* Of counter [1] (entries of the last seed scan's work list): those the
 * look-up's anchor-pair filter left without a seed -- a lower anchor of the same
 * probe is exact at the same window (the pair is reported from that anchor's
 * seed), or no second anchor of the probe matches (more than m mismatches).
 * They cost the verification 4 bytes each and none of its gathers. */
int catchhip_ctx_last_seeds_dropped(catchhip_ctx *ctx, int64_t *out);
/* Work of the row-parallel frontier solver in the last solve that used it (zeros
 * otherwise), 4 values: row records streamed by the count launches (alive rows,
 * summed over the rounds); of those, rows whose bitmap words were read again
 * because one of them had changed (SURVEY 8(d) K2's E_dirty -- the reference's
 * memo invalidation by overlap, catch/utils/set_cover.py:552-613, is the same
 * idea); bitmap words read for them; owner words the claim launches looked at. */
int catchhip_ctx_last_solver_counters(catchhip_ctx *ctx, int64_t *out4);
/* Work of the last Hamming near-duplicate filter on this context (SURVEY 8(d)
 * K3's quantities; catch/filter/near_duplicate_filter.py:47-142 does the same
 * look-ups probe by probe): probes N, tables T, pairs sharing a bucket that were
 * compared (C, all tables), pairs within the distance (edges). */
int catchhip_ctx_last_ndf_counters(catchhip_ctx *ctx, int64_t *out4);

/* ---- inputs ------------------------------------------------------------ */
/* Target sequences (catch/genome.py Genome.seqs of every genome of a group).
 * bytes: concatenation of the nseq sequences (raw characters, compared by
 * equality exactly as the reference compares str characters);
 * seq_off[nseq+1]; seq_genome[nseq] non-decreasing genome index in
 * [0, ngenomes). Uploads and bit-packs on the device. */
int catchhip_targets_create(catchhip_ctx *ctx, const uint8_t *bytes,
                            const int64_t *seq_off, const int32_t *seq_genome,
                            int64_t nseq, int32_t ngenomes,
                            catchhip_targets **out);
/* The same targets from one pointer per sequence (seq_ptr[i], seq_len[i] bytes
 * each; what a host that keeps its sequences as separate strings has): the
 * library gathers them into pinned memory with a few threads and uploads from
 * there, instead of the caller concatenating them first. */
int catchhip_targets_create_ptrs(catchhip_ctx *ctx, const uint8_t *const *seq_ptr,
                                 const int64_t *seq_len, const int32_t *seq_genome,
                                 int64_t nseq, int32_t ngenomes,
                                 catchhip_targets **out);
int catchhip_targets_destroy(catchhip_targets *t);
/* Hand-over of an input object to another context of the same device.  The
 * reference overlaps nothing here (catch/filter/set_cover_filter.py:816-846
 * builds a group's sets, then solves it); this path packs and uploads group
 * i + 1 on an upload context while the compute context scans and solves group
 * i, and the finished object changes hands: the call waits for the stream the
 * object was built on, after which only calls on `to` may use it. */
int catchhip_targets_rebind(catchhip_targets *t, catchhip_ctx *to);

/* Candidate probes + their seed ("anchor") table, i.e. the content of the
 * reference's kmer_probe_map (catch/probe.py:507-577 builds it,
 * :580-763 stores it): nprobes UNIQUE probe sequences (bytes/probe_off),
 * set_id[nprobes] = the set-cover set id that owns each unique probe
 * (catch/filter/set_cover_filter.py:408-412), and nent unique
 * (ent_probe, ent_pos) anchors of common length k. */
int catchhip_probes_create(catchhip_ctx *ctx, const uint8_t *bytes,
                           const int64_t *probe_off, int64_t nprobes,
                           const int32_t *set_id, const int32_t *ent_probe,
                           const int32_t *ent_pos, int64_t nent, int32_t k,
                           catchhip_probes **out);
int catchhip_probes_destroy(catchhip_probes *p);
int catchhip_probes_rebind(catchhip_probes *p, catchhip_ctx *to);   /* see catchhip_targets_rebind */

/* ---- K1: coverage scan -------------------------------------------------- */
#define CATCHHIP_SCAN_AUTO 0     /* seed scan when its preconditions hold, else general */
#define CATCHHIP_SCAN_GENERAL 1  /* force the seed-join + extension path */
#define CATCHHIP_SCAN_FAST 2     /* force the tiled Hamming kernel: the seed
                                    kernel's conditions + pigeonhole anchors
                                    {0,k,..,L-k} with L/k > mismatches (EINVAL
                                    otherwise) */
#define CATCHHIP_SCAN_SEED 3     /* force the hash-seeded Hamming kernel: equal-
                                    length DNA probes, lcf_thres == probe length,
                                    island == 0, sequences >= probe length; any
                                    anchor table (EINVAL otherwise) */
/* Replaces SetCoverFilter._make_sets (catch/filter/set_cover_filter.py
 * :359-470) = for every target sequence, probe.find_probe_covers_in_sequence
 * (catch/probe.py:1008-1271) under the default hybridization model
 * probe_covers_sequence_by_longest_common_substring(mismatches, lcf_thres,
 * island) (catch/probe.py:1274-1346, catch/utils/longest_common_substring.py
 * :59-159), then cover_extension / clip / genome offset (:424-453) and
 * IntervalSet normalisation (catch/utils/interval.py:25-44, :288-316).
 * Result: device-resident rows (set id, universe, start, end) sorted by
 * (set id, universe, start); *nrows = number of rows. */
int catchhip_cover_scan(catchhip_ctx *ctx, const catchhip_probes *probes,
                        const catchhip_targets *targets, int32_t mismatches,
                        int32_t lcf_thres, int32_t island,
                        int32_t cover_extension, int32_t mode,
                        catchhip_rows **out, int64_t *nrows);
/* The same scan without the interval merge: what coverage_analysis.Analyzer
 * ._find_covers_in_target_genomes collects (catch/coverage_analysis.py
 * :183-280) = for every sequence probe.find_probe_covers_in_sequence(
 * merge_overlapping=False) (catch/probe.py:1263-1270: duplicate ranges of a
 * probe removed, nothing merged), every range extended by cover_extension and
 * clipped to its sequence.  Rows (set id, universe, start, end) sorted by (set
 * id, start, end) in genome coordinates; pass one genome per sequence to keep
 * sequences apart.  EINVAL if one probe has more than 8192 ranges. */
int catchhip_cover_ranges(catchhip_ctx *ctx, const catchhip_probes *probes,
                          const catchhip_targets *targets, int32_t mismatches,
                          int32_t lcf_thres, int32_t island,
                          int32_t cover_extension, int32_t mode,
                          catchhip_rows **out, int64_t *nrows);
/* Copy rows to the host (arrays of length nrows). */
int catchhip_rows_fetch(catchhip_ctx *ctx, const catchhip_rows *rows,
                        int32_t *set_id, int32_t *universe, int64_t *start,
                        int64_t *end);
/* Build a rows object from host arrays (sorted by (set, universe, start),
 * disjoint non-touching within (set, universe)); genome_len[ngenomes] gives
 * the coordinate range of each universe.  Lets the greedy solver run on
 * externally supplied instances (the reference's set_cover unit tests). */
int catchhip_rows_from_host(catchhip_ctx *ctx, const int32_t *set_id,
                            const int32_t *universe, const int64_t *start,
                            const int64_t *end, int64_t nrows,
                            const int64_t *genome_len, int32_t ngenomes,
                            catchhip_rows **out);
int catchhip_rows_destroy(catchhip_rows *r);

/* Replaces SetCoverFilter._compute_tolerant_bp_covered_within_sequence
 * (catch/filter/set_cover_filter.py:472-529) for every sequence of
 * `targets` (the caller adds reverse-complement sequences to `targets`):
 * bp_out[i] += sum over sequences of the total length of the merged cover
 * ranges of unique probe i (bp_out has nprobes entries, host). */
int catchhip_tolerant_bp(catchhip_ctx *ctx, const catchhip_probes *probes,
                         const catchhip_targets *targets, int32_t mismatches,
                         int32_t lcf_thres, int32_t island, int64_t *bp_out);

/* ---- K2: greedy multi-universe set cover ------------------------------- */
/* Replaces set_cover.approx_multiuniverse(sets, costs == 1, universe_p,
 * ranks, use_intervalsets=True) (catch/utils/set_cover.py:147-615) as called
 * by catch/filter/set_cover_filter.py:113-144.  num_sets = number of
 * candidate probes (set ids 0..num_sets-1); ranks[num_sets] (NULL = all
 * equal); universe_p[ngenomes] float64 coverage fraction per universe
 * (NULL = 1.0).  out_ids (capacity num_sets) receives the chosen set ids,
 * *n_out their number, in the order in which the sequential algorithm picks
 * them (the reference returns a Python set; the order is the one of its loop).
 * When every universe_p is 1 and rows are at most 257 elements the solver takes
 * all locally-maximal sets per round (setcover_batched.inc) and restores that
 * order by sorting on the accept-time key; otherwise it picks one set per
 * iteration. */
int catchhip_setcover_greedy(catchhip_ctx *ctx, const catchhip_rows *rows,
                             int64_t num_sets, const int64_t *ranks,
                             const double *universe_p, int64_t *out_ids,
                             int64_t *n_out);

/* Replaces SetCoverFilter._filter's device work for one group in ONE call
 * (catch/filter/set_cover_filter.py:816-846 = _make_sets then
 * set_cover.approx_multiuniverse): catchhip_cover_scan + catchhip_setcover_
 * greedy with the rows never leaving the device.  Arguments as for those two
 * functions; *nrows (may be NULL) receives the number of cover rows. */
int catchhip_setcover_filter(catchhip_ctx *ctx, const catchhip_probes *probes,
                             const catchhip_targets *targets,
                             int32_t mismatches, int32_t lcf_thres,
                             int32_t island, int32_t cover_extension,
                             int32_t mode, int64_t num_sets,
                             const int64_t *ranks, const double *universe_p,
                             int64_t *out_ids, int64_t *n_out, int64_t *nrows);
/* The same for n independent groups at once (the reference handles the groups
 * of a design one after the other, set_cover_filter.py:816-846 per group):
 * group g runs on ctxs[g] -- its own HIP stream -- driven by its own host
 * thread, so the groups' kernels overlap on the device.  All arrays have n
 * entries; ranks[g] / universe_p[g] may be NULL, as may the two arrays
 * themselves; out_ids[g] has room for num_sets[g] ids.  Contexts must be
 * distinct.  Returns the first failing group's error code (its message through
 * catchhip_last_error of the calling thread). */
int catchhip_setcover_filter_many(int32_t n, catchhip_ctx *const *ctxs,
                                  const catchhip_probes *const *probes,
                                  const catchhip_targets *const *targets,
                                  int32_t mismatches, int32_t lcf_thres,
                                  int32_t island, int32_t cover_extension,
                                  int32_t mode, const int64_t *num_sets,
                                  const int64_t *const *ranks,
                                  const double *const *universe_p,
                                  int64_t *const *out_ids, int64_t *n_out,
                                  int64_t *nrows);

/* Multi-GPU, one process per GPU: an RCCL communicator attached to a context
 * (the reference has no counterpart: it forks a process pool,
 * catch/probe.py:727-743, catch/filter/set_cover_filter.py:848-900).  With a
 * communicator catchhip_setcover_greedy takes the per-pick form: every rank
 * holds the full rows, rank r evaluates the gains of sets s with s % nranks ==
 * r and the winner is agreed by one all-reduce(MAX) of a 64-bit key per pick. */
int catchhip_comm_unique_id(uint8_t *id128);
int catchhip_comm_init(catchhip_ctx *ctx, const uint8_t *id128, int32_t nranks,
                       int32_t rank);
int catchhip_comm_destroy(catchhip_ctx *ctx);
/* Collective: every rank of the communicator fills nelem uint32 with rank + 1, SUM
 * all-reduces them on the context's stream and checks every element against
 * nranks (nranks + 1) / 2.  CATCHHIP_ECOMM with the RCCL error or the number of wrong
 * elements.  catch_amd.parallel.init_from_env runs it once after catchhip_comm_init and
 * falls back to the host exchange, by agreement of all ranks, if any rank fails. */
int catchhip_comm_selftest(catchhip_ctx *ctx, int64_t nelem);
/* Which RCCL the communicators of this library go through: "RCCL version code V
 * from <file> (<how it was chosen>)".  The library opens ONE copy by path
 * (CATCHHIP_RCCL_PATH, else /opt/rocm/lib/librccl.so) instead of whatever file of
 * that soname the process mapped first (torch ships its own).  No reference
 * counterpart (the reference is single-process Python). */
int catchhip_comm_info(char *buf, int64_t len);

/* ONE set cover instance with its UNIVERSES (target genomes) sharded over
 * ranks -- set_cover.approx_multiuniverse (catch/utils/set_cover.py:147-615)
 * for a group too large for one GPU's share of the work.  Every rank scans
 * all candidate probes against its own contiguous range of the group's genomes
 * (catchhip_cover_scan on a targets object holding just those) and creates a
 * shard from its rows; num_sets and ranks are the same everywhere.  The
 * frontier solver then runs in rounds, each of which the caller drives:
 *     catchhip_shard_count        local gains           (enqueued)
 *     all-reduce SUM of the gain buffer   (catchhip_shard_allreduce(S, 0), or
 *                                          _allreduce_local for shards of one process)
 *     catchhip_shard_claim_check  claims + local losses (enqueued)
 *     all-reduce MAX of the lost buffer   (which = 1)
 *     catchhip_shard_apply        accepted sets applied; synchronises; *done =
 *                                 1 finished, -1 ranks exhausted, 0 next round
 * and catchhip_shard_picks returns the picks in the sequential pick order --
 * the same list on every rank, identical to the unsharded solver's.  Rows of
 * at most 257 bases; other instances are solved whole on one rank
 * (catch_amd/parallel.py).
 * Partial coverage (catchhip_shard_create_p with universe_p, one fraction per
 * LOCAL universe, some below 1; set_cover.py:362-373, 393-433): need[u] and the
 * acceptance thresholds of a universe live on the rank that owns it; a claimant
 * that lost nowhere must also pass the universe test on EVERY rank, so a round
 * has a third step between the lost exchange and the apply:
 *     catchhip_shard_verdict      local universe tests of the candidates left; a
 *                                 failure is one more lost mark (enqueued)
 *     all-reduce MAX of the lost buffer once more (which = 1)
 * (a no-op for shards created without universe_p).  Sharded by the row-parallel
 * kernels only (65,536 to 2^25 sets; CATCHHIP_EINVAL otherwise: solve it whole).
 * catchhip_shard_buffers exposes the two exchange buffers (device pointers:
 * uint32[gain_count], uint8[lost_count]) for callers with their own
 * transport.  Ask again before EVERY exchange: large shards pack their
 * buffers -- only the sets whose global gain was still positive after the
 * previous round (gains) / is positive in this one (marks) travel, in set
 * order, a list every rank derives from the same all-reduced gains -- so the
 * counts shrink from round to round (they are the same on every rank) and may
 * reach 0, in which case there is nothing to exchange. */
typedef struct catchhip_shard catchhip_shard;
int catchhip_shard_create(catchhip_ctx *ctx, const catchhip_rows *rows,
                          int64_t num_sets, const int64_t *ranks,
                          catchhip_shard **out);
int catchhip_shard_create_p(catchhip_ctx *ctx, const catchhip_rows *rows,
                            int64_t num_sets, const int64_t *ranks,
                            const double *universe_p, catchhip_shard **out);
/* instance_partial: 1 / 0 = whether any universe of the WHOLE instance (on any
 * rank) has a fraction below 1 -- the caller passes the same value on every
 * rank so that all of them run the same kernels and refuse the same instances
 * even when a rank's own universes all have fraction 1 (coverage given in
 * bases: catch/filter/set_cover_filter.py:761-792); -1 = decide from this
 * shard's universe_p (what catchhip_shard_create_p does). */
int catchhip_shard_create_pi(catchhip_ctx *ctx, const catchhip_rows *rows,
                             int64_t num_sets, const int64_t *ranks,
                             const double *universe_p, int instance_partial,
                             catchhip_shard **out);
int catchhip_shard_verdict(catchhip_shard *shard);
int catchhip_shard_destroy(catchhip_shard *shard);
int catchhip_shard_count(catchhip_shard *shard);
int catchhip_shard_claim_check(catchhip_shard *shard);
int catchhip_shard_apply(catchhip_shard *shard, int32_t *done);
int catchhip_shard_buffers(catchhip_shard *shard, void **gain,
                           int64_t *gain_count, void **lost,
                           int64_t *lost_count);
int catchhip_shard_picks(catchhip_shard *shard, int64_t *out_ids,
                         int64_t *n_out);
/* out4 = {elements of the gain buffer (uint32) and of the lost buffer (uint8)
 * in the NEXT exchange, bytes per lost element (1), 1 if the row-parallel
 * kernels of large shards run}. */
int catchhip_shard_info(catchhip_shard *shard, int64_t *out4);
/* which: 0 = gain buffer (SUM), 1 = lost buffer (MAX).  _allreduce uses the
 * context's RCCL communicator (stream-ordered); _allreduce_local reduces the
 * buffers of n shards that live in this process on one device. */
int catchhip_shard_allreduce(catchhip_shard *shard, int32_t which);
/* Exchange buffer `which` to (to_host != 0) or from a host array of the
 * buffer's size (synchronous): for a transport of the caller's own. */
int catchhip_shard_buffer_copy(catchhip_shard *shard, int32_t which, void *host,
                               int32_t to_host);
int catchhip_shard_allreduce_local(int32_t n, catchhip_shard *const *shards,
                                   int32_t which);
/* The whole round loop of a sharded instance in one call (round 6; catch_amd/parallel.py drove it from the
 * interpreter, one host synchronisation per round, until then): count -> all-reduce SUM -> claim -> all-reduce MAX ->
 * [partial coverage: verdict -> all-reduce MAX] -> apply, queued stream-ordered rounds_per_sync rounds at a time; the
 * exchange buffers keep the capacity of the last read-back, the done flag and the number of sets still alive are read
 * back once per batch.  shards: the n shards this process holds, fresh (round 0); transport 0: RCCL over the
 * context's communicator (one shard per process, one process per GPU: the production path, replaces the worker pool
 * of catch/filter/set_cover_filter.py:848-900 for a group that is sharded), 1: the shards of this process exchange
 * among themselves (one context; tests, and one-GPU runs of several ranges).  *done: 1 finished, -1 rank list
 * exhausted; catchhip_shard_picks then returns the picks. */
int catchhip_shard_solve(int32_t n, catchhip_shard *const *shards, int32_t transport, int32_t rounds_per_sync,
                         int32_t *done);

/* ---- K3: near-duplicate filter (Hamming LSH) --------------------------- */
/* Replaces NearDuplicateFilter._filter for NearDuplicateFilterWithHamming
 * Distance (catch/filter/near_duplicate_filter.py:47-142) with
 * lsh.NearNeighborLookup (catch/utils/lsh.py:239-320): n UNIQUE probes of
 * equal length L given in priority order (multiplicity descending, stable);
 * positions[ntables*k] the sampled positions per table; keep[n] (host,
 * uint8) receives 1 for probes that the sequential greedy pass includes. */
int catchhip_ndf_hamming(catchhip_ctx *ctx, const uint8_t *bytes, int64_t n,
                         int32_t L, const int32_t *positions, int32_t ntables,
                         int32_t k, int32_t dist_thres, uint8_t *keep);

/* Replaces NearDuplicateFilter._filter for NearDuplicateFilterWithMinHash
 * (catch/filter/near_duplicate_filter.py:148-190) with lsh.MinHashFamily(
 * kmer_size, N=1, use_fast_str_hash=True) and lsh.NearNeighborLookup
 * (catch/utils/lsh.py:48-215, :239-320).  The family's inner hash is the
 * interpreter's hash(str); this library computes it as CPython <= 3.10 does
 * under PYTHONHASHSEED=0 (SipHash-2-4, zero key) -- the only setting that
 * makes the reference's filter reproducible, and the one the golden vectors
 * were recorded with.  n UNIQUE probes in priority order (bytes / probe_off[n+1];
 * lengths may differ, each >= kmer_size <= 16, at most 256 k-mers per probe);
 * ab[ntables*k*2] = (a, b) of every hash function in the order the reference
 * draws them; a table key is the tuple of k minima of (a*|hash(kmer)|+b) mod
 * (2^31-1); two probes are near-duplicates when they share a key in some table
 * and the Jaccard distance of their k-mer sets (float64) is <= dist_thres. */
int catchhip_ndf_minhash(catchhip_ctx *ctx, const uint8_t *bytes,
                         const int64_t *probe_off, int64_t n, int32_t kmer_size,
                         const int64_t *ab, int32_t ntables, int32_t k,
                         double dist_thres, uint8_t *keep);

/* The same filter for `ngroups` independent groups of probes in one pass (the
 * clusters of --cluster-and-design-separately; BaseFilter.filter runs _filter
 * once per group, catch/filter/base_filter.py:111-175): group g = probes
 * [group_off[g], group_off[g+1]) in priority order, with its own hash functions
 * ab[g][ntables][k][2]; no probe is compared with a probe of another group. */
int catchhip_ndf_minhash_many(catchhip_ctx *ctx, const uint8_t *bytes,
                              const int64_t *probe_off, int64_t n,
                              const int64_t *group_off, int64_t ngroups,
                              int32_t kmer_size, const int64_t *ab,
                              int32_t ntables, int32_t k, double dist_thres,
                              uint8_t *keep);

/* ---- clustering pre-step (next row: catch/utils/cluster.py) --------------
 * MinHash signatures of sequences, replacing lsh.MinHashFamily(kmer_size, N)
 * .make_h() / h(s) (catch/utils/lsh.py:75-153) with the deterministic md5 inner
 * hash (:106-111) as cluster.make_signatures_with_minhash calls it
 * (catch/utils/cluster.py:28-44): per sequence the N smallest values, with
 * multiplicity, ascending, of (a * md5(kmer) + b) mod (2^31 - 1) over all
 * k-mers (md5 digest read as a big-endian 128-bit integer); a sequence with
 * fewer than N k-mers repeats them in whole rounds.  bytes = the sequences'
 * characters back to back (ASCII, as given -- no case folding), offsets[nseq+1]
 * their starts; 1 <= kmer_size <= 55 <= every sequence length; N <= 1024;
 * (a, b) as the reference draws them (1 <= a <= 2^31-1, 0 <= b <= 2^31-1). */
typedef struct catchhip_sigs catchhip_sigs;
int catchhip_sigs_create(catchhip_ctx *ctx, const uint8_t *bytes,
                         const uint64_t *offsets, uint32_t nseq,
                         int32_t kmer_size, uint32_t N, uint32_t a, uint32_t b,
                         catchhip_sigs **out);
/* The same with one pointer per sequence (the strings' own storage, e.g. CPython's ASCII str
 * buffers): gathered by host threads into pinned memory -- no multi-gigabyte join on the caller's side. */
int catchhip_sigs_create_ptrs(catchhip_ctx *ctx, const uint8_t *const *seq_ptr, const int64_t *seq_len,
                              uint32_t nseq, int32_t k, uint32_t N, uint32_t a, uint32_t b,
                              catchhip_sigs **out);
void catchhip_sigs_destroy(catchhip_sigs *sigs);
/* out[nseq * N]: signature of sequence s at out[s*N .. s*N+N) */
int catchhip_sigs_fetch(catchhip_ctx *ctx, const catchhip_sigs *sigs,
                        uint32_t *out);
/* MinHashFamily.estimate_jaccard_dist (catch/utils/lsh.py:170-215) of signature
 * j against every signature: common[k] = values the merge walk finds in both
 * (the walk always makes exactly N union steps, so the reference's distance is
 * 1.0 - common[k] / N in float64, which the caller forms). */
int catchhip_sigs_common_row(catchhip_ctx *ctx, const catchhip_sigs *sigs,
                             uint32_t j, uint16_t *common);
/* The same row reduced to the neighbours the connected-components search
 * (catch/utils/cluster.py:235-355) looks at: out[i] = (k << 16 | common[k]) for
 * every sequence k with common[k] >= min_common (its distance to j is within the
 * threshold), in no particular order; *count = how many (an error when it
 * exceeds cap).  Signatures of at most 176 values. */
int catchhip_sigs_neighbors(catchhip_ctx *ctx, const catchhip_sigs *sigs,
                            uint32_t j, uint32_t min_common,
                            unsigned long long *out, int64_t cap, int64_t *count);
/* The lists of up to 32 vertices in one launch (the search asks for the vertex it
 * explores together with those it has just stacked): out[i] = (q << 48 | k << 16 |
 * common[k]) for query q = js[q].  Signatures of at most 112 values.  *count beyond cap is
 * an error (CATCHHIP_EINVAL; nothing is written to out then). */
int catchhip_sigs_neighbors_many(catchhip_ctx *ctx, const catchhip_sigs *sigs,
                                 const uint32_t *js, int64_t nq, uint32_t min_common,
                                 unsigned long long *out, int64_t cap, int64_t *count);
/* The neighbour lists of ALL vertices at once (the graph the connected-components search of
 * catch/utils/cluster.py:235-355 walks): every ordered pair (j, k), j != k, with common(j, k) >= min_common.
 * Built and kept on the device; *nedges = how many.  max_edges > 0: when the graph has more, nothing is kept
 * (the caller falls back to catchhip_sigs_neighbors_many).  Signatures of at most 112 values. */
int catchhip_sigs_graph(catchhip_ctx *ctx, catchhip_sigs *sigs, uint32_t min_common,
                        int64_t max_edges, int64_t *nedges);
/* ... copied out in CSR form: the neighbours of j are idx[ptr[j] .. ptr[j + 1]) ascending, common[] beside them. */
int catchhip_sigs_graph_fetch(catchhip_ctx *ctx, const catchhip_sigs *sigs, int64_t *ptr,
                              uint32_t *idx, uint32_t *common);
/* The connected-components search of catch/utils/cluster.py:235-355 over such a graph, host side (no kernel):
 * runs the explored vertices whose neighbour order is known without building a Python set (catch_amd/utils/
 * cluster.py says when) and hands the others back.  ptr / idx / common are borrowed until catchhip_dfs_destroy;
 * a neighbour with common >= near_common is absorbed without being explored (the early-stop rule, :313-330).
 * catchhip_dfs_run(m = len(remaining)): *status 0 = finished; 1 = a component ended (catchhip_dfs_seen lists it;
 * remove it from `remaining`, run again); 2 = catchhip_dfs_set_copy_rank is needed (rank of every vertex in the
 * iteration order of remaining.copy()), run again; 3 = *vertex needs a real `remaining - queued`
 * (catchhip_dfs_new_queued lists what was queued since the last call; catchhip_dfs_push files the vertex's
 * neighbours in that set's order), run again.  catchhip_dfs_counts: explored vertices by case. */
typedef struct catchhip_dfs catchhip_dfs;
int catchhip_dfs_create(uint32_t n, const int64_t *ptr, const uint32_t *idx, const uint32_t *common,
                        uint32_t near_common, catchhip_dfs **out);
void catchhip_dfs_destroy(catchhip_dfs *dfs);
int catchhip_dfs_run(catchhip_dfs *dfs, int64_t m, int32_t *status, int64_t *vertex);
int catchhip_dfs_seen(catchhip_dfs *dfs, const uint32_t **p, int64_t *count);
int catchhip_dfs_new_queued(catchhip_dfs *dfs, const uint32_t **p, int64_t *count);
int catchhip_dfs_set_copy_rank(catchhip_dfs *dfs, const int64_t *rank);
/* the same from the members of remaining.copy() in its iteration order (a member's rank is its position) */
int catchhip_dfs_set_copy_members(catchhip_dfs *dfs, const int64_t *members, int64_t count);
int catchhip_dfs_push(catchhip_dfs *dfs, const int64_t *ks, const uint8_t *near, int64_t count);
int catchhip_dfs_counts(const catchhip_dfs *dfs, int64_t *out3);
/* The whole search in one call (round 6; nothing is handed back to the interpreter): `indices_to_consider`
 * (catch/utils/cluster.py:284, :343) is kept as CPython lays it out -- set(range(n)), `-=` with its dummies and
 * rebuilds, copy(), and the difference built insert by insert -- so the order of `list(a - b)` (:303-304) is the
 * interpreter's.  On a fresh handle.  comp[n]: vertices component after component in discovery order;
 * comp_ptr[n + 1]; stats[8] (may be null): explored vertices by the case of their difference (ascending, a copy of
 * the set, built insert by insert), how many of them had fewer than two new neighbours (no order to establish),
 * copies simulated, differences simulated, keys those inserted, orders read off the keys' home slots (the members
 * span fewer keys than the table has slots). */
int catchhip_dfs_run_all(catchhip_dfs *dfs, uint32_t *comp, int64_t *comp_ptr, int64_t *ncomp, int64_t *stats);
/* The emulated set by itself, for checking it against the interpreter (catch_amd/utils/cluster.py does so once per
 * process): create = set(range(n)); isub: s -= set(keys); list(which = 0: list(s), 1: list(s.copy()),
 * 2: list(s - set(keys))) -> *p (valid until the next call on the handle), *count. */
typedef struct catchhip_pyintset catchhip_pyintset;
int catchhip_pyintset_create(uint32_t n, catchhip_pyintset **out);
void catchhip_pyintset_destroy(catchhip_pyintset *s);
int catchhip_pyintset_isub(catchhip_pyintset *s, const uint32_t *keys, int64_t count);
int catchhip_pyintset_list(catchhip_pyintset *s, int32_t which, const uint32_t *keys, int64_t count,
                           const uint32_t **p, int64_t *n);
/* cluster.create_condensed_dist_matrix (catch/utils/cluster.py:102-194) for the
 * signature distance: out[n(n-1)/2] float32 in SciPy's condensed order, entry
 * (i, j) = lut[common(i, j)] with lut[N+1] supplied by the caller (the float32
 * roundings of 1.0 - c / N, which is what the reference's c_float array holds). */
int catchhip_sigs_condensed(catchhip_ctx *ctx, const catchhip_sigs *sigs,
                            const float *lut, float *out);

/* ---- scan with first-discovery keys (next row: catch/filter/adapter_filter.py)
 * catchhip_cover_scan, plus for every row the key that orders probes the way
 * the reference's result dict does: find_probe_covers_in_sequence walks a
 * sequence left to right, looks every k-mer up in the k-mer -> {(probe, pos)}
 * map and inserts a probe into its result when the first of its seeds is
 * accepted (catch/probe.py:1062-1108, :1253-1271); AdapterFilter's interval
 * scheduling breaks ties between equal range ends by that insertion order
 * (catch/filter/adapter_filter.py:191-240, catch/utils/interval.py:319-358).
 * Per (set, universe) group: first_key = (position of that first accepted k-mer
 * relative to the universe's start) << 32 | anchor_order[entry], where
 * anchor_order[nanchors] (may be null = 0) is the caller's rank of each anchor
 * entry inside its k-mer's entry list (the reference iterates a Python set
 * there; the host mirrors it).  The probes must have been created with their
 * anchors sorted by (probe, position) without duplicates; give every sequence
 * its own universe to get per-sequence keys.  Not available with
 * CATCHHIP_SCAN_FAST. */
int catchhip_cover_scan_first_seen(catchhip_ctx *ctx, const catchhip_probes *probes,
                                   const catchhip_targets *targets,
                                   int32_t mismatches, int32_t lcf_thres,
                                   int32_t island_of_exact_match,
                                   int32_t cover_extension, int32_t mode,
                                   const uint32_t *anchor_order,
                                   catchhip_rows **out, int64_t *nrows);
/* first_key[nrows], aligned with catchhip_rows_fetch */
int catchhip_rows_fetch_first_seen(catchhip_ctx *ctx, const catchhip_rows *rows,
                                   uint64_t *first_key);

/* ---- independent instances in one scan ------------------------------------
 * The groups of a design are independent set cover instances
 * (catch/filter/set_cover_filter.py:816-846: _make_sets per group, one solve
 * per group).  Many small groups (the clusters of
 * --cluster-and-design-separately) can share one probes / targets pair: give
 * every probe and every genome the number of its group; the scans then report
 * a probe only inside genomes of its own group.  The rows of the groups are
 * disjoint in sets and universes, so one greedy solve over all of them makes,
 * restricted to a group, exactly that group's own sequence of picks (a set's
 * gain depends only on its own group's universes; ties go to the lowest id).
 * Pass null to remove the groups. */
int catchhip_probes_set_groups(catchhip_ctx *ctx, catchhip_probes *probes,
                               const int32_t *group_of_probe);
int catchhip_targets_set_groups(catchhip_ctx *ctx, catchhip_targets *targets,
                                const int32_t *group_of_genome);

/* ---- candidate probes on the device (front end, next row #1) ---------------
 * candidate_probes.make_candidate_probes_from_sequences over all sequences of a
 * targets object, genome by genome (catch/filter/candidate_probes.py:21-182 as
 * ProbeDesigner calls it, catch/filter/probe_designer.py:250-262: windows of
 * probe_length every probe_stride bases, the window flush with the sequence end
 * when the length is not a multiple of the stride, no window holding two or
 * more consecutive 'N', and the windows flanking every run of >= 2 'N'),
 * followed by DuplicateFilter (catch/filter/duplicate_filter.py:16-26: first
 * occurrences, order kept).  The unique candidates stay on the device as
 * positions in the targets; their index in first-occurrence order is the set id
 * the set cover works with.  seq_length_to_skip < 0 = none; a sequence shorter
 * than probe_length that is not skipped is an error, as in the reference
 * (--small-seq-min is handled by the host's string path).  `targets` must
 * outlive the candidates object. */
typedef struct catchhip_candidates catchhip_candidates;
int catchhip_candidates_create(catchhip_ctx *ctx, const catchhip_targets *targets,
                               int32_t probe_length, int32_t probe_stride,
                               int64_t seq_length_to_skip,
                               catchhip_candidates **out, int64_t *ncandidates,
                               int64_t *nunique);
void catchhip_candidates_destroy(catchhip_candidates *cands);
/* Iteration order of a CPython set.  The reference's near-duplicate filters
 * return `list(to_include)` -- a set of Probe objects hashed by hash(seq_str)
 * (catch/filter/near_duplicate_filter.py:76-103, catch/probe.py:324-329) -- and
 * SetCoverFilter numbers its candidates in the order it receives them, so which of
 * two equally good probes is picked follows from that order.  order[i] = index of
 * the i-th key a set iterates after the n distinct keys with these hashes were added
 * in index order (Objects/setobject.c of CPython 3.7-3.12).  ..._strs hashes the
 * strings as CPython <= 3.10 does under PYTHONHASHSEED=0 (SipHash-2-4, zero key) --
 * the one setting under which the reference's own order is reproducible; the
 * catchhip_candidates_ndf_* calls leave their candidates in this order.
 * Host-only integer work (no device involved). */
int catchhip_pyset_order(const int64_t *hashes, int64_t n, int64_t *order);
int catchhip_pyset_order_strs(const uint8_t *bytes, const int64_t *off, int64_t n,
                              int64_t *order);
/* The same order computed on the device (what the catchhip_candidates_ndf_* calls use for
 * more than a few thousand kept probes): all insertions of one table generation at once,
 * a slot claimed with atomicMin on (insertion rank, index) words and the displaced key
 * walking on -- the fixed point is the sequential table.  hashes / order: host arrays. */
int catchhip_pyset_order_device(catchhip_ctx *ctx, const int64_t *hashes, int64_t n, int64_t *order);
int catchhip_candidates_rebind(catchhip_candidates *cands, catchhip_ctx *to);   /* see catchhip_targets_rebind */
/* global_start[i] = position (in the targets' concatenated coordinate) of the
 * first occurrence of unique candidate ids[i] (ids == NULL: candidates 0..n-1) */
int catchhip_candidates_fetch(catchhip_ctx *ctx, const catchhip_candidates *cands,
                              const int64_t *ids, int64_t n, int64_t *global_start);
/* NearDuplicateFilter._filter on the candidates (catch/filter/
 * near_duplicate_filter.py:47-103; arguments as catchhip_ndf_hamming /
 * catchhip_ndf_minhash): the unique candidates are taken in the filters'
 * priority order -- multiplicity among ALL candidates descending, ties in
 * first-occurrence order (:60-66) -- and the object's list becomes the kept
 * ones in that order, which is the order the set cover filter then numbers
 * them in.  At most one such call per candidates object. */
int catchhip_candidates_ndf_hamming(catchhip_ctx *ctx, catchhip_candidates *cands,
                                    const int32_t *positions, int32_t ntables,
                                    int32_t k, int32_t dist_thres, int64_t *nkept);
int catchhip_candidates_ndf_minhash(catchhip_ctx *ctx, catchhip_candidates *cands,
                                    int32_t kmer_size, const int64_t *ab,
                                    int32_t ntables, int32_t k, double dist_thres,
                                    int64_t *nkept);
/* Grouped targets (catchhip_targets_set_groups before catchhip_candidates_create):
 * duplicates are only removed inside a group, the candidates of a group stay
 * together (groups in order), the priority order is per group, and
 * catchhip_probes_from_candidates passes the groups on to the probes.  The
 * near-duplicate filters then take one set of sampled positions / hash
 * functions per group (ab[ngroups][ntables][k][2] as catchhip_ndf_minhash_many);
 * catchhip_candidates_groups returns the group of every unique candidate. */
int catchhip_candidates_ndf_hamming_many(catchhip_ctx *ctx, catchhip_candidates *cands,
                                         const int32_t *positions /* [ngroups][ntables][k] */,
                                         int64_t ngroups, int32_t ntables, int32_t k,
                                         int32_t dist_thres, int64_t *nkept);
int catchhip_candidates_ndf_minhash_many(catchhip_ctx *ctx, catchhip_candidates *cands,
                                         int32_t kmer_size, const int64_t *ab,
                                         int64_t ngroups, int32_t ntables, int32_t k,
                                         double dist_thres, int64_t *nkept);
int catchhip_candidates_groups(catchhip_ctx *ctx, const catchhip_candidates *cands,
                               int32_t *group_of_candidate);
/* A probes object of the unique candidates (set id = candidate index), as
 * catchhip_probes_create would build from their strings.  Anchors: sorted by
 * (probe, position) without duplicates, or ent_probe = ent_pos = NULL for the
 * pigeonhole table {0, k, 2k, ..} of every probe (k must divide probe_length). */
int catchhip_probes_from_candidates(catchhip_ctx *ctx, const catchhip_candidates *cands,
                                    const int32_t *ent_probe, const int32_t *ent_pos,
                                    int64_t nent, int32_t k, catchhip_probes **out);
/* The same with RANDOM anchors given as their draws (catch/probe.py:356-405:
 * construct_kmer_probe_map_to_find_probe_covers draws num_kmers_per_probe = 20
 * positions per probe with np.random, repeats allowed; the map keeps a probe
 * once per distinct k-mer position): draws[probe][draws_per_probe], one byte
 * each (probe_length - k + 1 <= 256), in the order np.random.randint made them.
 * The table is every probe's sorted distinct positions, built on the device. */
int catchhip_probes_from_candidates_draws(catchhip_ctx *ctx, const catchhip_candidates *cands,
                                          const uint8_t *draws, int32_t draws_per_probe,
                                          int32_t k, catchhip_probes **out);

/* Statistics of a row table without fetching it (coverage analysis of large
 * designs, catch/coverage_analysis.py:282-335): per universe total_len = sum of
 * (end - start) over its rows (:318-320) and union_len = bases covered by at
 * least one row (:282-302); per set id < num_sets the number of universes it
 * has rows in (probe_map_counts, :255-258, with one universe per sequence).
 * Every set id in the table must be < num_sets when num_sets > 0. */
int catchhip_rows_stats(catchhip_ctx *ctx, const catchhip_rows *rows,
                        int64_t *total_len, int64_t *union_len,
                        int64_t num_sets, int64_t *universes_per_set);

/* Property checks of a set-cover solution by kernels independent of the solvers (csrc/check.hip), usable at any
 * scale: `picks` are set ids IN THE ORDER they were picked, `rows` the instance's row table (not a deferred one),
 * universe_p per universe or NULL (every universe fully covered).  out5[0] = picks that covered no new position
 * when their turn came (catch/utils/set_cover.py:448-550 only picks a set with a positive gain), out5[1] =
 * universes covered short of |U| - int(|U| - p |U|) (set_cover.py:362-373, U = union of all rows of the universe),
 * out5[2] = pick ids out of range or repeated, out5[3] = bases in the universes, out5[4] = of them covered.
 * A correct solution has out5[0] == out5[1] == out5[2] == 0. */
int catchhip_rows_cover_check(catchhip_ctx *ctx, const catchhip_rows *rows, int64_t num_sets, const int64_t *picks,
                              int64_t npicks, const double *universe_p, int64_t *out5);

/* AdapterFilter._make_votes_across_target_genomes (catch/filter/adapter_filter.py
 * :299-361) on rows from catchhip_cover_scan_first_seen with one universe per
 * sequence: per sequence, in order, the greedy interval schedule over its
 * ranges sorted by end (ties by first-discovery key; catch/utils/interval.py
 * :319-358) gives the scheduled probes an 'A' vote and the other hybridizing
 * probes a 'B' vote, and the sequence's votes are swapped when that makes the
 * sum over all probes of max(A, B) strictly larger; multiplicity[set id] = how
 * many input probes equal that probe (each counts in the sums).  Returns the
 * totals per set id (num_sets must exceed every set id of the table). */
int catchhip_adapter_votes(catchhip_ctx *ctx, const catchhip_rows *rows,
                           int64_t num_sets, const int64_t *multiplicity,
                           int64_t *votes_a, int64_t *votes_b);

#ifdef __cplusplus
}
#endif
#endif /* CATCHHIP_H */
