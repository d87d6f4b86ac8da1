"""Where does an end-to-end design run spend its time?  (GPU box)
    python tools/e2e_profile.py S3 0.2
Writes the groups of a synthetic dataset as FASTA files, then times the stages
of `catch_amd.design` (FASTA -> candidates -> duplicate filter -> set cover)."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catch_amd import design  # noqa: E402
from catch_amd.utils import synthetic  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "S2"
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    extra = sys.argv[3:]
    groups = synthetic.dataset(name, scale=scale)
    tmp = tempfile.mkdtemp()
    files = []
    for gi, genomes in enumerate(groups):
        fn = os.path.join(tmp, "g%d.fasta" % gi)
        with open(fn, "w") as f:
            for j, g in enumerate(genomes):
                for c, s in enumerate(g):
                    f.write(">g%d_%d_%d\n%s\n" % (gi, j, c, s))
        files.append(fn)
    out = os.path.join(tmp, "probes.fasta")
    argv = files + ["-o", out, "-pl", "100", "-ps", "50", "-m", "2", "-e", "50"] + extra
    args = design.parse_args(argv)
    design.main(design.parse_args(argv))      # warm-up (library load, allocator)
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    design.main(args)
    pr.disable()
    el = time.perf_counter() - t0
    print("end-to-end %.3f s for %d groups, %d genomes" % (el, len(groups), sum(len(g) for g in groups)))
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
