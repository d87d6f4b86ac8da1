"""Timeline of one M2 pass of the bench's S4 workload (host strings in, ids out): pack / build per item, scan + solve
per lane.   python tools/m2_timeline.py [workload] [scale]"""
import os, sys, time
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import OrderedDict
from catch_amd import genome
from catch_amd.filter.set_cover_filter import SetCoverFilter
from catch_amd.utils import synthetic

wl = sys.argv[1] if len(sys.argv) > 1 else "S4"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
groups = synthetic.dataset(wl, scale=scale)
gobjs = [[genome.Genome(list(g)) if len(g) == 1 else genome.Genome.from_chrs(OrderedDict((str(i), s) for i, s in enumerate(g)))
          for g in grp] for grp in groups]
f = SetCoverFilter(mismatches=2, lcf_thres=100, coverage=1.0, cover_extension=50)
for rep in range(4):
    t = time.perf_counter()
    f._filter_genomes_device(gobjs, 100, 50, return_ids=True)
    dt = time.perf_counter() - t
    print("pass %d: %.4f s" % (rep, dt))
sizes = [sum(len(s) for g in grp for s in g) for grp in groups]
for ev in sorted(f.last_timings["pipe_events"], key=lambda e: e[2]):
    print("   %-14s item %3d (%6.1f Mb)  %.4f -> %.4f  (%.4f)" % (ev[0], ev[1], sizes[ev[1]] / 1e6, ev[2], ev[3], ev[3] - ev[2]))
