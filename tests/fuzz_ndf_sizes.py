"""Randomised stress of the near-duplicate filters at the sizes where the lazy resolution compacts its tables (> 4,096
probes; GPU box):   python tests/fuzz_ndf_sizes.py [seconds] [seed]   |   python tests/fuzz_ndf_sizes.py big [probes] [seed]
Families of near-identical sequences (strains of a few species) next to unrelated ones, 4 k - 80 k candidate probes,
with and without groups; both LSH families.  The default form (wake-ups, queue, probe passes, packed states, compaction)
must keep exactly what the polling rounds of round 3 keep (CATCHHIP_NDF_POLL_ROUNDS=1) -- two implementations of the same
fixed point that share only the comparison kernels -- and, for small cases, what the oracle keeps."""
import os
os.environ.setdefault("CATCHHIP_TEST_HOOKS", "1")
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catch_amd.filter import candidate_probes  # noqa: E402
from catch_amd.filter.near_duplicate_filter import (NearDuplicateFilterWithHammingDistance,  # noqa: E402
                                                    NearDuplicateFilterWithMinHash)
from oracle import oracle  # noqa: E402


def sequences(rnd):
    seqs = []
    for _ in range(rnd.randrange(1, 6)):                       # species
        root = "".join(rnd.choice("ACGT") for _ in range(rnd.randrange(800, 6000)))
        rate = rnd.choice([0.0, 0.002, 0.01, 0.03, 0.08])
        for _ in range(rnd.randrange(2, 120)):                 # strains
            s = list(root)
            for i in range(len(s)):
                if rnd.random() < rate:
                    s[i] = rnd.choice("ACGT")
            if rnd.random() < 0.1:
                i = rnd.randrange(len(s))
                s[i] = "N"
            seqs.append("".join(s))
    return seqs


def one_case(seed):
    rnd = random.Random(seed)
    seqs = sequences(rnd)
    stride = rnd.choice([10, 25, 50])
    cands = candidate_probes.candidate_strings_from_sequences(seqs, 100, stride)
    if rnd.random() < 0.5:
        cands = list(dict.fromkeys(cands))
    if len(cands) > 80000:
        cands = cands[:80000]
    ngroups = rnd.choice([1, 1, 3, 17])
    cuts = sorted(rnd.sample(range(1, len(cands)), ngroups - 1)) if ngroups > 1 and len(cands) > ngroups else []
    parts = [cands[a:b] for a, b in zip([0] + cuts, cuts + [len(cands)])]
    mh_d, ham_d = rnd.choice([0.3, 0.5, 0.6, 0.8]), rnd.choice([1, 2, 4, 8])

    def run():
        random.seed(seed)
        a = NearDuplicateFilterWithMinHash(mh_d)._filter_strs_many(parts)
        random.seed(seed + 1)
        b = [NearDuplicateFilterWithHammingDistance(ham_d, 100)._filter_strs(p) for p in parts]
        return a, b
    os.environ.pop("CATCHHIP_NDF_POLL_ROUNDS", None)
    got = run()
    os.environ["CATCHHIP_NDF_POLL_ROUNDS"] = "1"
    try:
        want = run()
    finally:
        os.environ.pop("CATCHHIP_NDF_POLL_ROUNDS", None)
    assert got == want, ("forms differ", seed, len(cands), ngroups, mh_d, ham_d)
    if len(cands) <= 6000 and ngroups == 1:
        random.seed(seed)
        f = NearDuplicateFilterWithMinHash(mh_d)
        params = f._draw_params()
        assert sorted(got[0][0]) == sorted(oracle.ndf_minhash(cands, mh_d, params)), ("oracle", seed)
    return len(cands)


def big_case(nprobes, seed):
    """ONE case at the size of an S5 chunk (default 5 M candidate probes: what the control loop of the lazy resolution
    sees in configs[4] -- tables compacted several times, the deferred-walk queue in use, thousands of parked probes):
    strains of a few species generated with numpy, both LSH families, the default form against the polling rounds."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    want_bases = nprobes * 50
    seqs, bases = [], 0
    while bases < want_bases:
        n = int(rng.integers(60000, 400000))
        root = rng.integers(0, 4, size=n, dtype=np.uint8)
        rate = float(rng.choice([0.002, 0.01, 0.03, 0.08]))
        for _ in range(int(rng.integers(20, 300))):
            s = root.copy()
            m = rng.random(n) < rate
            s[m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
            seqs.append(lut[s].tobytes().decode())
            bases += n
            if bases >= want_bases:
                break
    cands = candidate_probes.candidate_strings_from_sequences(seqs, 100, 50)
    del seqs
    ngroups = 1 + seed % 3
    cuts = sorted(int(c) for c in rng.choice(np.arange(1, len(cands)), size=ngroups - 1, replace=False)) if ngroups > 1 else []
    parts = [cands[a:b] for a, b in zip([0] + cuts, cuts + [len(cands)])]

    def run():
        t0 = time.time()
        random.seed(seed)
        a = NearDuplicateFilterWithMinHash(0.6)._filter_strs_many(parts)
        t1 = time.time()
        random.seed(seed + 1)
        b = [NearDuplicateFilterWithHammingDistance(2, 100)._filter_strs(p) for p in parts]
        return a, b, t1 - t0, time.time() - t1
    os.environ.pop("CATCHHIP_NDF_POLL_ROUNDS", None)
    got = run()
    os.environ["CATCHHIP_NDF_POLL_ROUNDS"] = "1"
    try:
        want = run()
    finally:
        os.environ.pop("CATCHHIP_NDF_POLL_ROUNDS", None)
    assert got[0] == want[0], ("MinHash forms differ", seed, len(cands))
    assert got[1] == want[1], ("Hamming forms differ", seed, len(cands))
    print("fuzz_ndf_sizes big: %d probes in %d group(s), seed %d: MinHash keeps %d (%.1f s; polling rounds %.1f s), "
          "Hamming keeps %d (%.1f s; %.1f s) -- both forms equal"
          % (len(cands), ngroups, seed, sum(map(len, got[0])), got[2], want[2], sum(map(len, got[1])), got[3], want[3]), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        big_case(int(sys.argv[2]) if len(sys.argv) > 2 else 5000000, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
        return
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle.build()
    t0, n, tot = time.time(), 0, 0
    while time.time() - t0 < budget:
        try:
            tot += one_case(seed0 + n)
        except BaseException:
            print("fuzz_ndf_sizes: FAILED at seed %d" % (seed0 + n), flush=True)
            raise
        n += 1
    print("fuzz_ndf_sizes: %d cases ok (seeds %d..%d, %d probes in all) in %.0f s" % (n, seed0, seed0 + n - 1, tot, time.time() - t0))


if __name__ == "__main__":
    main()
