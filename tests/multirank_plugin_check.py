"""Run under torch.distributed.run by tests/test_gpu_parity.py: the SetCoverFilter
plugin over several ranks (whole groups by LPT + one group sharded by
universes) must select, on EVERY rank, exactly what the oracle selects --
with the pigeonhole anchors and with random anchors (np.random stream)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

from catch_amd import genome, parallel, probe  # noqa: E402
from catch_amd.filter.set_cover_filter import SetCoverFilter  # noqa: E402
from catch_amd.utils import synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402
sys.path.insert(0, os.path.join(REPO, "tests"))
from util import candidates  # noqa: E402


def main():
    W = parallel.init_from_env()
    assert W.size > 1
    orc.build()
    rng = np.random.Generator(np.random.PCG64(5))
    groups = [synthetic.make_species(rng, [2600], 4, 2, 0.05, 0.01),
              synthetic.make_species(rng, [5200], 14, 3, 0.06, 0.012),     # the big one: sharded
              synthetic.make_species(rng, [1800], 3, 1, 0.0, 0.02),
              synthetic.make_species(rng, [2100], 5, 2, 0.04, 0.01)]
    cands = [candidates(g, 100, 50) for g in groups]
    gen = [[genome.Genome.from_one_seq(g[0]) for g in grp] for grp in groups]
    ok = True
    for m in (2, 5):
        np.random.seed(31)
        exp = orc.set_cover_filter(cands, groups, m, 100, coverage=1.0, cover_extension=50)
        np.random.seed(31)
        f = SetCoverFilter(mismatches=m, lcf_thres=100, coverage=1.0, cover_extension=50)
        out = f.filter([[probe.Probe.from_str(s) for s in c] for c in cands], gen,
                       input_is_grouped=True)
        got = [sorted(p.seq_str for p in g) for g in out]
        want = [sorted(c[i] for i in ids) for c, ids in zip(cands, exp)]
        ok = ok and got == want and f.last_timings.get("sharded_groups") == [1]
    # --identify: the sharded group's ranks (a tolerant scan of its candidates over ALL groups and their reverse
    # complements) are computed on every rank and gate the sharded rounds' claims
    exp = orc.set_cover_filter(cands, groups, 2, 100, coverage=1.0, cover_extension=50, identify=True)
    f = SetCoverFilter(mismatches=2, lcf_thres=100, coverage=1.0, cover_extension=50, identify=True)
    out = f.filter([[probe.Probe.from_str(s) for s in c] for c in cands], gen, input_is_grouped=True)
    got = [sorted(p.seq_str for p in g) for g in out]
    want = [sorted(c[i] for i in ids) for c, ids in zip(cands, exp)]
    ok = ok and got == want and f.last_timings.get("sharded_groups") == [1]
    # -c 0.9 (round 4): the sharded group's universes keep their needs and acceptance thresholds on the rank that
    # owns them; with the row-parallel kernels forced (an instance this small would take the set-parallel ones,
    # which refuse partial coverage: then the group is solved whole -- both ways the oracle's selection)
    exp = orc.set_cover_filter(cands, groups, 2, 100, coverage=0.9, cover_extension=50)
    want = [sorted(c[i] for i in ids) for c, ids in zip(cands, exp)]
    for forced in ("1", "0"):
        os.environ["CATCHHIP_SHARD_FLAT"] = forced
        f = SetCoverFilter(mismatches=2, lcf_thres=100, coverage=0.9, cover_extension=50)
        out = f.filter([[probe.Probe.from_str(s) for s in c] for c in cands], gen, input_is_grouped=True)
        got = [sorted(p.seq_str for p in g) for g in out]
        ok = ok and got == want and f.last_timings.get("sharded_groups") == [1]
    # coverage given in BASES: the sharded group's first nine genomes are shorter than the
    # coverage asked for (p == 1.0), the last five longer (p < 1): with two ranks the first rank's own universes are
    # all full-coverage ones -- it must still build a partial shard and run the partial rounds (the instance is)
    groups_b = [g for g in groups]
    groups_b[1] = [[g[0][:4000]] for g in groups[1][:9]] + [list(g) for g in groups[1][9:]]
    cands_b = [candidates(g, 100, 50) for g in groups_b]
    gen_b = [[genome.Genome.from_one_seq(g[0]) for g in grp] for grp in groups_b]
    exp = orc.set_cover_filter(cands_b, groups_b, 2, 100, coverage=4500, cover_extension=50)
    want = [sorted(c[i] for i in ids) for c, ids in zip(cands_b, exp)]
    os.environ["CATCHHIP_SHARD_FLAT"] = "1"
    f = SetCoverFilter(mismatches=2, lcf_thres=100, coverage=4500, cover_extension=50)
    out = f.filter([[probe.Probe.from_str(s) for s in c] for c in cands_b], gen_b, input_is_grouped=True)
    got = [sorted(p.seq_str for p in g) for g in out]
    ok = ok and got == want and f.last_timings.get("sharded_groups") == [1]
    os.environ.pop("CATCHHIP_SHARD_FLAT", None)
    res = W.allgather(ok)
    if W.rank == 0:
        print("MULTIRANK_PLUGIN_OK" if all(res) else "MULTIRANK_PLUGIN_MISMATCH %s" % res)
    W.close()


if __name__ == "__main__":
    main()
