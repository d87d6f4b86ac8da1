"""Times of the neighbour graph of the clustering step on S5 x scale (GPU box): device pass, copy, edges."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")
import numpy as np
from catch_amd import engine
from catch_amd.utils import lsh, synthetic
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
genomes = synthetic.dataset("S5", scale=scale)[0]
seqs = []
for g in genomes:
    for s in g:
        seqs.extend(s[i:i + 50000] for i in range(0, len(s), 50000))
random.seed(6)
fam = lsh.MinHashFamily(12, N=100)
t0 = time.perf_counter()
sigs = fam.signatures(seqs)
print("signatures of %d fragments: %.3f s" % (len(seqs), time.perf_counter() - t0))
ctx = sigs.ctx
for rep in range(2):
    t0 = time.perf_counter()
    cnt = engine.ctypes.c_int64(0)
    engine.check(ctx._L.catchhip_sigs_graph(ctx._h, sigs._h, 9, 0, engine.ctypes.byref(cnt)))
    t1 = time.perf_counter()
    print("graph build: %.3f s wall, device %.1f ms, %d ordered pairs" % (t1 - t0, ctx.kernel_ms(engine.PHASE_NDF)[0], cnt.value))
    t0 = time.perf_counter()
    g = sigs.graph(9)
    print("graph() incl. fetch: %.3f s" % (time.perf_counter() - t0))
deg = np.diff(g[0])
print("degree: mean %.1f max %d, vertices without neighbours %d" % (deg.mean(), deg.max(), int((deg == 0).sum())))
