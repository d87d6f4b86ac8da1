"""GPU box: cProfile of the bench's resident S4 step (host time outside the C call)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from catch_amd import engine
from catch_amd.utils import synthetic
groups = synthetic.dataset("S4")
st = bench.Stepper(0, groups, list(range(len(groups))), 4)
for _ in range(2):
    st.step()
st.sync()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    stats = []
    st.step(stats)
st.sync()
pr.disable()
print("ms per step", (time.perf_counter() - t0) / 5 * 1e3)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3500])
