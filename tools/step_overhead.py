"""Where a bench step spends host time outside the C call (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from catch_amd import engine

groups, cands = bench.make_workload("S2", 2, 1.0)
ctxs = [engine.Context(0) for _ in groups]
res = [bench.ResidentGroup(c, g, cd) for c, g, cd in zip(ctxs, groups, cands)]
specs = [(g.ctx, g.probes, g.targets, g.n_sets, None, None) for g in res]
for _ in range(20):
    engine.setcover_filter_many(specs, 2, 100, 0, 50)
N = 300
t0 = time.perf_counter()
for _ in range(N):
    engine.setcover_filter_many(specs, 2, 100, 0, 50)
t1 = time.perf_counter()
print("engine.setcover_filter_many: %.1f us per call" % ((t1 - t0) / N * 1e6))
ctxs[0].has_comm = False
stats = dict(scan_ms=0.0, rows_ms=0.0, greedy_ms=0.0, scan_launches=0, greedy_launches=0, picks=0, rows=0)
t0 = time.perf_counter()
for _ in range(N):
    bench.one_step(ctxs[0], res, stats, 0)
t1 = time.perf_counter()
print("bench.one_step with stats:   %.1f us per call" % ((t1 - t0) / N * 1e6))
# the raw C call with prebuilt arguments
import ctypes, numpy as np
from catch_amd._lib import c_i64p, c_f64p
L = ctxs[0]._L
n = len(specs)
VP = ctypes.c_void_p
cx = (VP * n)(*[g[0]._h for g in specs]); pr = (VP * n)(*[g[1]._h for g in specs]); tg = (VP * n)(*[g[2]._h for g in specs])
nsets = np.array([g[3] for g in specs], dtype=np.int64)
outs = [np.empty(g[3], dtype=np.int64) for g in specs]
out_p = (c_i64p * n)(*[o.ctypes.data_as(c_i64p) for o in outs])
rk = (c_i64p * n)(); up = (c_f64p * n)()
n_out = np.zeros(n, dtype=np.int64); nrows = np.zeros(n, dtype=np.int64)
t0 = time.perf_counter()
for _ in range(N):
    L.catchhip_setcover_filter_many(n, cx, pr, tg, 2, 100, 0, 50, 0, nsets.ctypes.data_as(c_i64p), rk, up, out_p,
                                    n_out.ctypes.data_as(c_i64p), nrows.ctypes.data_as(c_i64p))
t1 = time.perf_counter()
print("raw C call:                  %.1f us per call" % ((t1 - t0) / N * 1e6))
