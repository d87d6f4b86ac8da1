"""Design aid (not product code): simulate the batched greedy round structure
on the CPU to see how many rounds different candidate rules need.

    python tests/sim_batched_rounds.py [dataset] [group]

Rule "full": every live set is a candidate (theoretical minimum of rounds for
the word-granular ownership test).  Rule "topT": per-thread top-T keys over
1024 threads, horizon = max (T+1)-th key (what gb_select_kernel does, T=1).
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle  # noqa: E402
from catch_amd.utils import synthetic  # noqa: E402
from tests.util import candidates  # noqa: E402


def simulate(sid, gstart, gend, nsets, total, rule, T=1, nthreads=1024,
             gran=64):
    unc = np.ones(total, dtype=bool)
    picked = np.zeros(nsets, dtype=bool)
    rounds = []
    while True:
        cs = np.concatenate([[0], np.cumsum(unc)])
        rc = cs[gend] - cs[gstart]
        gain = np.bincount(sid, weights=rc, minlength=nsets).astype(np.int64)
        gain[picked] = 0
        if gain.max() == 0:
            break
        key = (gain << 24) | (0xFFFFFF - np.arange(nsets))
        key[gain == 0] = 0
        if rule == "full":
            cand = gain > 0
        else:
            H = 0
            for t in range(nthreads):
                ks = np.sort(key[t::nthreads])[::-1]
                if len(ks) > T:
                    H = max(H, ks[T])
            cand = key > H
        live = cand[sid] & (rc > 0)
        idx = np.nonzero(live)[0]
        owner = np.zeros(total // gran + 2, dtype=np.int64)
        # word-granular claims (only words with uncovered bits of the row)
        for i in idx:
            a, b = gstart[i], gend[i]
            for w in range(a // gran, (b - 1) // gran + 1):
                lo, hi = max(a, w * gran), min(b, (w + 1) * gran)
                if cs[hi] - cs[lo] > 0 and key[sid[i]] > owner[w]:
                    owner[w] = key[sid[i]]
        acc = cand.copy()
        for i in idx:
            a, b = gstart[i], gend[i]
            for w in range(a // gran, (b - 1) // gran + 1):
                lo, hi = max(a, w * gran), min(b, (w + 1) * gran)
                if cs[hi] - cs[lo] > 0 and owner[w] != key[sid[i]]:
                    acc[sid[i]] = False
        acc &= gain > 0
        for i in np.nonzero(acc[sid])[0]:
            unc[gstart[i]:gend[i]] = False
        picked |= acc
        rounds.append((int(cand.sum()), int(live.sum()), int(acc.sum())))
    return rounds


def main():
    ds = sys.argv[1] if len(sys.argv) > 1 else "S2"
    gi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    genomes = synthetic.dataset(ds)[gi]
    cand = candidates(genomes, 100, 50)
    k, entries = oracle.anchor_table(cand, 2, 100)
    uniq, owner = oracle._unique_last(cand)
    pr, un, st, en = oracle.make_sets(uniq, entries, k, genomes, 2, 100, 0, 50)
    own = np.array(owner, dtype=np.int64)
    sid = own[pr]
    glen = np.array([sum(len(s) for s in g) for g in genomes])
    base = np.concatenate([[0], np.cumsum(glen)])
    gs, ge = base[un] + st, base[un] + en
    total = int(base[-1])
    print("rows", len(sid), "sets", len(cand), "total", total)
    for rule, T in (("top", 1), ("top", 4), ("full", 0)):
        for gran in (64, 1):
            r = simulate(sid, gs, ge, len(cand), total, rule, T, gran=gran)
            print(rule, T, "gran", gran, "rounds", len(r), "picks",
                  sum(x[2] for x in r))
            print("   ", r[:12], "...")


if __name__ == "__main__":
    main()
