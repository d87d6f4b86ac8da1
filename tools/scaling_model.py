#!/usr/bin/env python3
"""Expected S4 step at 1 / 2 / 4 / 8 GPUs -- a MODEL, written down before an 8-GPU node has run the code, so that
the first measured curve (SCALE_rNN.json) has something to be compared with.

    tools/scaling_model.py --measure > profiles/r04_scaling_inputs.json     (GPU box, one MI355X)
    tools/scaling_model.py profiles/r04_scaling_inputs.json                 (anywhere: prints the table)

Inputs measured on ONE GPU (resident inputs, as bench.py's step):
  * per group g: t_whole[g] = one fused scan + solve (catchhip_setcover_filter), ms;
  * for the group(s) the plan shards (catch_amd.parallel.plan_with_sharding: S4's 265-Mbase group from 3 ranks up)
    and every N: the shards' work done one after the other on the one device -- scan of all candidates against
    1/N of the genomes (t_scan[N][r]), and the sharded solve's launches of all N shards (t_rounds[N], so one
    rank's share is t_rounds[N] / N), the number of rounds and the packed exchange sizes round by round.
Model, per N (bench.py --gpus N: ONE dataset, strong scaling):
  rank r:  sum of t_whole over its whole groups (LPT plan)  +  for every sharded group:
           t_scan[N][r] + t_rounds[N] / N + rounds x 2 x (ALPHA + host sync) + exchange bytes x 2 (N-1)/N / BW
  step = max over ranks.
ALPHA (all-reduce latency), BW (per-link bandwidth a ring all-reduce sustains over xGMI) and the host read-back
per exchange are ASSUMPTIONS, stated in the output; MI355X_MICROARCH.md gives 7 links x ~153 GB/s per GPU, a ring
all-reduce is per-link bound.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")

ALPHA_US = 25.0          # one small RCCL all-reduce over xGMI, launch to completion (assumed)
SYNC_US = 30.0           # one 4-byte read-back + stream synchronisation per exchange of the packed form (assumed)
BW_GBS = 100.0           # what a ring all-reduce sustains per link (assumed: ~2/3 of the 153 GB/s link peak)
NS = (2, 4, 8)
RPS = 4                  # rounds queued per host read-back (catchhip_shard_solve)


def measure():
    import numpy as np   # noqa: F401
    from catch_amd import engine, parallel, probe
    from catch_amd.utils import synthetic
    groups = synthetic.dataset("S4")
    bases = [sum(len(s) for g in grp for s in g) for grp in groups]
    ctx = engine.default_context()
    out = {"workload": "S4", "bases": bases, "t_whole_ms": [], "sharded": {}}
    for gi, genomes in enumerate(groups):
        t = engine.Targets(ctx, genomes)
        c = engine.Candidates(ctx, t, 100, 50)
        k, ep, eo = probe.anchor_entries_equal_length(c.n, 100, 2, 100)
        p = c.probes(k, ep, eo)
        best = None
        for _ in range(3):
            ctx.sync()
            t0 = time.perf_counter()
            engine.setcover_filter(ctx, p, t, 2, 100, 0, 50, c.n)
            ctx.sync()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        out["t_whole_ms"].append(best)
        p.close(); c.close(); t.close()
    want = sorted({gi for n in NS for gi in parallel.plan_with_sharding(bases, n, min_cost=30_000_000)[0]})
    for gi in want:
        genomes = groups[gi]
        full = engine.Targets(ctx, genomes)
        cands = engine.Candidates(ctx, full, 100, 50)
        k, ep, eo = probe.anchor_entries_equal_length(cands.n, 100, 2, 100)
        probes = cands.probes(k, ep, eo)
        rec = {}
        for n in NS:
            b = parallel.split_universes([sum(len(s) for s in g) for g in genomes], n)
            shards, scans = [], []
            for r in range(n):
                t = engine.Targets(ctx, genomes[b[r]:b[r + 1]])
                best = None
                rows = None
                for _ in range(2):
                    if rows is not None:
                        rows.close()
                    ctx.sync()
                    t0 = time.perf_counter()
                    rows = engine.Rows.scan(ctx, probes, t, 2, 100, 0, 50, 0)
                    ctx.sync()
                    dt = (time.perf_counter() - t0) * 1e3
                    best = dt if best is None else min(best, dt)
                scans.append(best)
                shards.append(engine.Shard(rows, cands.n))
            gains, marks, rnd = [], [], 0
            ctx.sync()
            t0 = time.perf_counter()
            while True:
                for sh in shards:
                    sh.count()
                gains.append(shards[0]._exchange_shape()[0])
                engine.shards_allreduce_local(shards, 0)
                for sh in shards:
                    sh.claim_check()
                marks.append(shards[0]._exchange_shape()[1])
                engine.shards_allreduce_local(shards, 1)
                done = [sh.apply() for sh in shards]
                rnd += 1
                if done[0]:
                    break
            ctx.sync()
            t_py = (time.perf_counter() - t0) * 1e3
            for sh in shards:
                sh.close()
            # the same solve with the round loop under the C ABI (round 6: catchhip_shard_solve, RPS rounds per host
            # read-back, exchange buffers at the capacity of the last read-back); fresh shards over the same rows
            shards = [engine.Shard(sh.rows, cands.n) for sh in shards]
            ctx.sync()
            t0 = time.perf_counter()
            engine.shards_solve(shards, "local", RPS)
            ctx.sync()
            t_c = (time.perf_counter() - t0) * 1e3
            cap_g, cap_m = [], []
            for r in range(rnd):
                cap_g.append(gains[r - r % RPS])                       # capacity = the alive sets at the last read-back (+ 2)
                cap_m.append(max(gains[r - r % RPS] - 2, 0))
            rec[str(n)] = {"t_scan_ms": scans, "t_rounds_all_shards_ms": t_c, "t_rounds_all_shards_python_loop_ms": t_py,
                           "rounds": rnd, "rounds_per_sync": RPS, "gain_elements": cap_g, "mark_elements": cap_m,
                           "gain_elements_exact": gains, "mark_elements_exact": marks, "sets": int(cands.n)}
            for sh in shards:
                sh.close()
        out["sharded"][str(gi)] = rec
        probes.close(); cands.close(); full.close()
    json.dump(out, sys.stdout, indent=1)


def model(path):
    from catch_amd import parallel
    with open(path) as f:
        d = json.load(f)
    bases, tw = d["bases"], d["t_whole_ms"]
    print("assumptions: all-reduce latency %.0f us, host read-back %.0f us (one per exchange in round 4's inputs, one per "
          "rounds_per_sync rounds since round 6), ring bandwidth %.0f GB/s per link" % (ALPHA_US, SYNC_US, BW_GBS))
    print("| GPUs | sharded groups | busiest rank: whole groups ms | + sharded scan ms | + rounds ms | + exchange ms | step ms | speed-up | efficiency |")
    print("|---|---|---|---|---|---|---|---|---|")
    t1 = sum(tw)
    rows = [(1, [], t1, 0.0, 0.0, 0.0, t1)]
    for n in NS:
        sharded, plan = parallel.plan_with_sharding(bases, n, min_cost=30_000_000)
        best = None
        for r in range(n):
            whole = sum(tw[g] for g in plan[r])
            scan = rounds = exch = 0.0
            for g in sharded:
                s = d["sharded"][str(g)][str(n)]
                scan += s["t_scan_ms"][r]
                rounds += s["t_rounds_all_shards_ms"] / n
                nbytes = 4.0 * sum(s["gain_elements"]) + 1.0 * sum(s["mark_elements"])
                rps = s.get("rounds_per_sync")          # (round 4's inputs: a read-back per exchange)
                syncs = s["rounds"] * 2 if rps is None else -(-s["rounds"] // rps)
                exch += ((s["rounds"] * 2 * ALPHA_US + syncs * SYNC_US) * 1e-3
                         + nbytes * 2.0 * (n - 1) / n / (BW_GBS * 1e9) * 1e3)
            tot = whole + scan + rounds + exch
            if best is None or tot > best[-1]:
                best = (n, sharded, whole, scan, rounds, exch, tot)
        rows.append(best)
    for n, sharded, whole, scan, rounds, exch, tot in rows:
        print("| %d | %s | %.1f | %.1f | %.1f | %.1f | %.1f | %.2f | %.0f %% |"
              % (n, sharded or "-", whole, scan, rounds, exch, tot, t1 / tot, 100.0 * t1 / tot / n))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--measure":
        measure()
    else:
        model(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                 "profiles", "r04_scaling_inputs.json"))
