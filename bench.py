#!/usr/bin/env python3
"""Benchmark of the SetCoverFilter hot path (K1 coverage scan + row build +
K2 greedy set cover) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one full pass of the hot path over the workload with the packed
inputs already resident in HBM: one catchhip_setcover_filter_many call = for
every group (each on its own stream) the hash-seeded coverage scan, the
bucketed row build and the frontier set-cover solver.
Workload at N=1: BASELINE.json configs[1] -- ~100 Ebola+Lassa-like genomes,
`design.py -pl 100 -ps 50 -m 2 -e 50` -- as the seeded synthetic set S2
(catch_amd/utils/synthetic.py; the reference ships no input at this scale).
For N>1 (launched by torch.distributed.run, one rank per GPU) every rank
processes its own S2-shaped dataset (seed 2+rank): groups are independent
set-cover instances, so they shard with no data-path collective (weak
scaling).  `--shard probes` instead runs ONE dataset on all ranks with the
candidate sets sharded and one RCCL all-reduce(MAX) per greedy pick.

Prints one JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

from catch_amd import engine, probe  # noqa: E402
from catch_amd.filter import candidate_probes  # noqa: E402
from catch_amd.utils import synthetic  # noqa: E402

PROBE_LEN, STRIDE, MISMATCHES, EXT = 100, 50, 2, 50
SCAN_MODE = int(os.environ.get("CATCHHIP_SCAN_MODE", "0"))   # 0 auto, 1 general, 2 fast
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec


def make_workload(name, seed, scale):
    groups = synthetic.dataset(name, seed=seed, scale=scale)
    cands = []
    for genomes in groups:
        c = []
        for g in genomes:
            c += [p.seq_str for p in
                  candidate_probes.make_candidate_probes_from_sequences(
                      list(g), probe_length=PROBE_LEN, probe_stride=STRIDE)]
        cands.append(list(dict.fromkeys(c)))     # DuplicateFilter
    return groups, cands


class ResidentGroup:
    def __init__(self, ctx, genomes, cand):
        self.ctx = ctx
        self.n_sets = len(cand)
        self.G = sum(len(s) for g in genomes for s in g)
        k, uniq, owner, ep, eo = probe.anchor_table(cand, MISMATCHES, PROBE_LEN)
        self.targets = engine.Targets(ctx, genomes)
        self.probes = engine.Probes(ctx, uniq, owner, ep, eo, k)
        self.n_unique = len(uniq)

    def close(self):
        self.probes.close()
        self.targets.close()


def one_step(ctx, groups, stats=None, in_flight=0):
    """One pass of the hot path over every group.  in_flight = 0: all groups
    at once (catchhip_setcover_filter_many: one stream + host thread per
    group); 1: one group after the other (catchhip_setcover_filter)."""
    if ctx.has_comm:
        # probe-sharded RCCL solver: separate scan / solve calls on one context
        res = []
        for g in groups:
            rows = engine.Rows.scan(ctx, g.probes, g.targets, MISMATCHES,
                                    PROBE_LEN, 0, EXT, SCAN_MODE)
            res.append((rows.greedy(g.n_sets), rows.n))
            rows.close()
    else:
        specs = [(g.ctx, g.probes, g.targets, g.n_sets, None, None)
                 for g in groups]
        if in_flight == 1:
            res = [engine.setcover_filter(*sp[:3], MISMATCHES, PROBE_LEN, 0, EXT,
                                          sp[3], mode=SCAN_MODE) for sp in specs]
        else:
            res = engine.setcover_filter_many(specs, MISMATCHES, PROBE_LEN, 0,
                                              EXT, SCAN_MODE)
    if stats is not None:
        for g, (ids, nrows) in zip(groups, res):
            c = g.ctx
            ms, nl = c.kernel_ms(engine.PHASE_SCAN)
            stats["scan_ms"] += ms
            stats["scan_launches"] += nl
            stats["rows_ms"] += c.kernel_ms(engine.PHASE_ROWS)[0]
            stats["rows"] += nrows
            ms, nl = c.kernel_ms(engine.PHASE_GREEDY)
            stats["greedy_ms"] += ms
            stats["greedy_launches"] += nl
            rms, rnl = c.kernel_ms(engine.PHASE_GREEDY_ROUNDS)
            stats["rounds_ms"] = stats.get("rounds_ms", 0.0) + rms
            stats["rounds_launches"] = stats.get("rounds_launches", 0) + rnl
            stats["picks"] += len(ids)
            cn = c.counters()
            for k in ("raw_hits", "seed_hits", "winner_rows", "rows_recounted",
                      "bitmap_words_read", "greedy_iters"):
                stats[k] = stats.get(k, 0) + cn[k]
    return [ids for ids, _ in res]


def pmc_traffic(unit, workload, scale):
    """HBM bytes per launch of `unit` (a kernel, or the pair of kernels of a
    solver round) from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in
    separate runs of this same command; FETCH_SIZE doubled for gfx950 as
    MI355X_MICROARCH.md prescribes).  None when no matching record exists:
    counters cannot be collected from inside the timed run."""
    path = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec.get("workload") != workload or scale != 1.0:
            return None
        k = rec["units"][unit]
        return (2.0 * k["FETCH_SIZE_KB_per_launch"]
                + k["WRITE_SIZE_KB_per_launch"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(groups, cands, budget_s=12.0):
    """The CPU oracle (oracle/, plain C, 1 thread) timed on the same
    workload; reported beside the GPU number, never part of it."""
    from oracle import oracle as orc
    orc.build()
    units = sum(len(c) * sum(len(s) for g in grp for s in g)
                for c, grp in zip(cands, groups))
    t0 = time.perf_counter()
    reps = 0
    sel = None
    while True:
        sel = orc.set_cover_filter(cands, groups, MISMATCHES, PROBE_LEN,
                                   coverage=1.0, cover_extension=EXT)
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 50:
            break
    return dict(value=units * reps / el, unit="probe*bp/s", cores=1,
                kind="port", seconds_per_pass=el / reps,
                sample="%d full passes of the bench workload through the "
                       "plain-C oracle (seed-and-extend scan + interval-set "
                       "greedy), single thread" % reps), sel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="S2")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--shard", choices=["groups", "probes"], default="groups")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--groups-in-flight", type=int, default=0,
                    help="0 = all groups of the batch at once (one stream "
                         "each), 1 = one after the other")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist   # plumbing only: rendezvous/barrier
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on stdout; rank 0 must print ONE line
        import ctypes
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)

    ndev = max(1, engine.device_count())
    ctx = engine.Context(local_rank % ndev)   # one rank per GPU on a real node
    seed = 2 + (rank if args.shard == "groups" else 0)
    groups, cands = make_workload(args.workload, seed, args.scale)
    if args.shard == "probes":
        ids = [engine.Context.comm_unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(ids, src=0)
        ctx.comm_init(ids[0], world, rank)

    t_up0 = time.perf_counter()
    # one context (= one HIP stream) per group, so that independent groups can
    # be in flight together; the probe-sharded mode keeps everything on ctx
    ctx.has_comm = args.shard == "probes"
    gctx = [ctx if (i == 0 or ctx.has_comm) else engine.Context(ctx.device)
            for i in range(len(groups))]
    resident = [ResidentGroup(c, g, cd) for c, g, cd in zip(gctx, groups, cands)]
    for c in gctx:
        c.sync()
    upload_s = time.perf_counter() - t_up0
    units = sum(r.n_sets * r.G for r in resident)

    def barrier():
        for c in gctx:
            c.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        one_step(ctx, resident, None, args.groups_in_flight)
    stats = dict(scan_ms=0.0, rows_ms=0.0, greedy_ms=0.0, scan_launches=0,
                 greedy_launches=0, picks=0, rows=0)
    barrier()
    t0 = time.perf_counter()
    picks = None
    for _ in range(args.steps):
        picks = one_step(ctx, resident, stats, args.groups_in_flight)
    for c in gctx:
        c.sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        u = torch.tensor([float(units)], dtype=torch.float64)
        if args.shard == "groups":
            dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total_units = float(u[0])
        dist.barrier()
    else:
        total_units = float(units)

    if rank == 0:
        K = args.steps
        P = sum(r.n_sets for r in resident)
        G = sum(r.G for r in resident)
        rows_per_step = stats["rows"] / K
        # Algorithmic bytes per step (DESIGN.md "Roofline accounting"):
        #  K1 seed scan: 2 planes of the targets (0.25 B/base) + one hash-table
        #    probe per position (8 B key + 8 B range) + per seed: work-list
        #    write+read (24 B), the target window (3 planes x 5 words = 60 B),
        #    the probe image (64 B) and the hit record (20 B)
        #  rows: per hit 20 B record read + 12 B scatter + 12 B sort read; per
        #    merged row 12 B write + 12 B read + 16 B final row
        #  K2 round: per re-counted row 8 B record + 1 B flag, per bitmap word
        #    8 B read + 8 B owner word, per set and round 12 B of state
        seeds_step = stats.get("seed_hits", 0) / K
        hits_step = stats.get("raw_hits", 0) / K
        k1_bytes = 16.25 * G + 168.0 * seeds_step
        rows_bytes = 44.0 * hits_step + 40.0 * rows_per_step
        k1_ms = stats["scan_ms"] / K
        rows_ms = stats["rows_ms"] / K
        k1_launch_ms = stats["scan_ms"] / max(stats["scan_launches"], 1)
        k1_gbs = k1_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        rows_gbs = rows_bytes / (rows_ms * 1e-3) / 1e9 if rows_ms > 0 else 0.0
        picks_per_step = stats["picks"] / K
        k2_ms = stats["greedy_ms"] / K
        rounds_step = stats.get("greedy_iters", 0) / K
        k2_bytes_step = (9.0 * stats.get("rows_recounted", 0)
                         + 16.0 * stats.get("bitmap_words_read", 0)) / K \
            + 12.0 * P * rounds_step
        k2_rounds_ms = stats.get("rounds_ms", 0.0) / K     # the round launches only
        k2_launches = stats.get("rounds_launches", 0)
        k2_gbs = k2_bytes_step / (k2_rounds_ms * 1e-3) / 1e9 if k2_rounds_ms > 0 else 0.0
        # dominant kernel = the unit with the most device time per step.  A
        # solver round is one unit: gf_count_claim_kernel + gf_check_apply_kernel
        # (two dependent launches; HIP events bracket the whole batch of rounds)
        phases = {"k1_seed_scan": k1_ms, "rows_build": rows_ms,
                  "k2_solver_rounds": k2_rounds_ms}
        dominant = max(phases, key=phases.get)
        if dominant == "k2_solver_rounds":
            pair_ms = stats.get("rounds_ms", 0.0) / max(k2_launches // 2, 1)
            roof = dict(bound="hbm",
                        kernel="gf_count_claim_kernel+gf_check_apply_kernel",
                        achieved=k2_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=k2_gbs / HBM_PEAK_GBS,
                        traffic=pmc_traffic("solver_round", args.workload, args.scale),
                        algorithmic_bytes_per_launch=(
                            k2_bytes_step * K / max(k2_launches // 2, 1)),
                        avg_launch_ms=pair_ms,
                        launches_are="pairs (count+claim, check+apply), "
                                     "including the no-op pairs after the last round",
                        us_per_pick=k2_rounds_ms * 1e3 / max(picks_per_step, 1),
                        rounds_per_step=rounds_step)
        elif dominant == "k1_seed_scan":
            roof = dict(bound="hbm", kernel="seed scan (6 launches)",
                        achieved=k1_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=k1_gbs / HBM_PEAK_GBS,
                        traffic=pmc_traffic("seed_scan", args.workload, args.scale),
                        algorithmic_bytes_per_launch=k1_bytes * K / max(stats["scan_launches"], 1),
                        avg_launch_ms=k1_launch_ms)
        else:
            roof = dict(bound="hbm", kernel="bucketed row build",
                        achieved=rows_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=rows_gbs / HBM_PEAK_GBS,
                        traffic=pmc_traffic("rows_build", args.workload, args.scale),
                        algorithmic_bytes_per_launch=rows_bytes / 6.0,
                        avg_launch_ms=rows_ms / 6.0)
        out = {
            "metric": "candidate-probe x target-bp / s through SetCoverFilter "
                      "(K1 scan + K2 greedy)",
            "value": total_units * K / elapsed,
            "unit": "probe*bp/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if args.shard == "groups" else "strong",
            "vs_baseline": None,
            "dtype": "u32 bit-planes / u64 bitmap (integer)",
            "data": "synthetic",
            "config": {"workload": "%s (BASELINE configs[%d]%s): %d genomes in "
                                   "%d groups, G=%d bp, P=%d candidates, "
                                   "-pl 100 -ps 50 -m 2 -e 50 -c 1.0"
                                   % (args.workload,
                                      {"S1": 0, "S2": 1, "S3": 2, "S4": 3}.get(args.workload, 1),
                                      "" if args.scale == 1.0 else " scaled x%g" % args.scale,
                                      sum(len(g) for g in groups),
                                      len(groups), G, P),
                       "per_rank": True, "shard": args.shard,
                       "groups_in_flight": (len(groups) if args.groups_in_flight == 0
                                            else args.groups_in_flight),
                       "scale": args.scale},
            "setcoverfilter_ms": elapsed / K * 1e3,
            "picks": picks_per_step, "rows": rows_per_step,
            "kernel_ms_per_step": {"k1_scan": k1_ms,
                                   "rows_build": rows_ms,
                                   "k2_greedy": k2_ms,
                                   "k2_greedy_rounds_only": k2_rounds_ms},
            "k1_probe_bp_per_s": (sum(r.n_unique * r.G for r in resident)
                                  / (k1_ms * 1e-3)) if k1_ms > 0 else None,
            "roofline": roof,
            "roofline_k1": dict(bound="hbm", achieved=k1_gbs,
                                peak=HBM_PEAK_GBS, unit="GB/s",
                                frac=k1_gbs / HBM_PEAK_GBS,
                                avg_launch_ms=k1_launch_ms),
            "roofline_rows": dict(bound="hbm", achieved=rows_gbs,
                                  peak=HBM_PEAK_GBS, unit="GB/s",
                                  frac=rows_gbs / HBM_PEAK_GBS),
            "roofline_k2": dict(bound="hbm", achieved=k2_gbs,
                                peak=HBM_PEAK_GBS, unit="GB/s",
                                frac=k2_gbs / HBM_PEAK_GBS,
                                us_per_pick=k2_rounds_ms * 1e3 / max(picks_per_step, 1),
                                rounds_per_step=rounds_step),
            "work_per_step": {"seeds": seeds_step, "hits": hits_step,
                              "rows": rows_per_step,
                              "rows_recounted": stats.get("rows_recounted", 0) / K,
                              "bitmap_words_read": stats.get("bitmap_words_read", 0) / K},
            "h2d_upload_s": upload_s,
            "value_incl_h2d": total_units / (elapsed / K + upload_s),
        }
        if world == 1 and not args.no_cpu_baseline:
            base, sel = cpu_baseline(groups, cands)
            out["cpu_baseline"] = base
            # the timed GPU result must equal the oracle's (parity guard)
            out["parity_vs_oracle"] = [sorted(a) for a in picks] == \
                [sorted(b) for b in sel]
            out["speedup_vs_cpu_oracle"] = out["value"] / base["value"]
        print(json.dumps(out))
    for r in resident:
        r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
