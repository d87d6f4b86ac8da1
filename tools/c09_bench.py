"""S4 with partial coverage (-c 0.9): per group the fused filter with universe_p = 0.9, digests against
tests/golden/full_size_picks.json (picks_c09_*), device ms per phase.  GPU box."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from catch_amd import engine, probe
from catch_amd.utils import synthetic

cov = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
gold = {g["group"]: g for g in json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                           "tests", "golden", "full_size_picks.json")))["S4"]["groups"]}
groups = synthetic.dataset("S4")
ctx = engine.default_context()
tot = dict(wall=0.0, scan=0.0, rows=0.0, greedy=0.0, picks=0, rounds=0)
ok = True
for gi, genomes in enumerate(groups):
    if only is not None and gi not in only:
        continue
    t = engine.Targets(ctx, genomes)
    c = engine.Candidates(ctx, t, 100, 50)
    k, ep, eo = probe.anchor_entries_equal_length(c.n, 100, 2, 100)
    p = c.probes(k, ep, eo)
    up = [cov] * len(genomes)
    for rep in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        ids, nrows = engine.setcover_filter(ctx, p, t, 2, 100, 0, 50, c.n, universe_p=up)
        ctx.sync()
        wall = time.perf_counter() - t0
    g = gold[gi]
    key = "picks_c09" if cov == 0.9 else "picks"
    a = np.sort(np.asarray(ids, dtype=np.int64))
    same = (len(ids) == g["n_" + key] and hashlib.sha256(a.astype("<i8").tobytes()).hexdigest() == g[key + "_sha256"]
            and hashlib.sha256(np.asarray(ids, dtype="<i8").tobytes()).hexdigest() == g[key + "_in_order_sha256"])
    ok = ok and same
    cn = ctx.counters()
    ms = {n: ctx.kernel_ms(ph)[0] for n, ph in (("scan", engine.PHASE_SCAN), ("rows", engine.PHASE_ROWS), ("greedy", engine.PHASE_GREEDY))}
    print("group %2d rows %9d picks %6d rounds %6d wall %8.2f ms scan %7.2f rows %7.2f greedy %8.2f  digest %s"
          % (gi, nrows, len(ids), cn["greedy_iters"], wall * 1e3, ms["scan"], ms["rows"], ms["greedy"], "ok" if same else "DIFFERENT"))
    tot["wall"] += wall; tot["scan"] += ms["scan"]; tot["rows"] += ms["rows"]; tot["greedy"] += ms["greedy"]
    tot["picks"] += len(ids); tot["rounds"] += cn["greedy_iters"]
    p.close(); c.close(); t.close()
print(json.dumps(dict(coverage=cov, digests_ok=ok, **tot)))
