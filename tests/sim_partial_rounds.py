"""CPU simulation of frontier rounds under PARTIAL coverage (universe_p < 1): the acceptance rule the
row-parallel solver uses (setcover_flat.inc, partial mode), checked against the oracle's sequential picks.

A set c is accepted in a round iff
  (i)  it holds the largest key (gain, ~id) on every bitmap word where it has uncovered bits, and
  (ii) for every universe u it touches:  K(c) >= T_u  or  K(c) >= M_u,  where
         M_u = largest owner key of a word of u that holds uncovered bits of u,
         T_u = smallest owner key k of u with  weight(words of u owned by keys > k) <= need[u] - B
               where B bounds what c covers in u: nothing picked before c's turn can then bring need[u]
               below c's count in u, so neither min(need, count) nor its neighbours change c's gain before
               c is the maximum.  B = the largest (set, universe) count until round 5; since round 6 the
               smallest of the levels 8, 16, 32, 64, 128, CMAX that is >= c's own count in u when LEVELS=1
               is set (gr_usel / gr_passes built with -DGR_UT_LEVELS=6; exact, measured, not the default).
Gains are exact: sum over universes of min(need[u], count).  Usage: python tests/sim_partial_rounds.py [scale] [groups]
"""
import sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
from catch_amd.filter import candidate_probes
from catch_amd.utils import synthetic
from oracle import oracle as orc


import os
LEVELS = os.environ.get("LEVELS", "0") != "0"     # (the library builds with one level, GR_UT_LEVELS; LEVELS=1: six)


def level_of(c, bounds):
    for li, b in enumerate(bounds):
        if c <= b:
            return li
    return len(bounds) - 1


def simulate(rs, ru, gs, ge, goff, P, p):
    U = len(goff) - 1
    G = int(goff[-1])
    diff = np.zeros(G + 1, dtype=np.int64); np.add.at(diff, gs, 1); np.add.at(diff, ge, -1)
    unc = np.cumsum(diff[:-1]) > 0
    usize = np.add.reduceat(unc.astype(np.int64), goff[:-1]); usize[goff[1:] == goff[:-1]] = 0
    can = (usize - p * usize).astype(np.int64)
    need = usize - can
    for u in np.nonzero(need <= 0)[0]:
        unc[goff[u]:goff[u + 1]] = False
    segkey = rs.astype(np.int64) * U + ru
    segstart = np.concatenate(([0], np.nonzero(np.diff(segkey))[0] + 1))
    seg_set, seg_u = rs[segstart], ru[segstart]
    setstart = np.concatenate(([0], np.nonzero(np.diff(seg_set))[0] + 1))
    set_ids = seg_set[setstart].astype(np.int64)
    seg_of_set = np.searchsorted(set_ids, seg_set)          # dense set index of every segment
    row_seg = np.repeat(np.arange(len(segstart)), np.diff(np.concatenate((segstart, [len(rs)]))))
    cmax = int(np.add.reduceat(ge - gs, segstart).max())
    nwords = G // 64 + 2
    word_u = np.searchsorted(goff, np.arange(nwords) * 64, side="right") - 1   # universe of a word's first base
    picks, rounds, per_round = [], 0, []
    alive_set = np.ones(len(set_ids), dtype=bool)
    while (need > 0).any():
        rounds += 1
        cs = np.concatenate(([0], np.cumsum(unc)))
        cnt = cs[ge] - cs[gs]
        segcnt = np.add.reduceat(cnt, segstart)
        contrib = np.minimum(segcnt, np.maximum(need[seg_u], 0))
        gain = np.add.reduceat(contrib, setstart)
        gain[~alive_set] = 0
        key = gain.astype(np.int64) * (1 << 32) + ((1 << 32) - 1 - set_ids)
        key[gain == 0] = 0
        # owner per word: max key of a set with an uncovered bit in the word
        owner = np.zeros(nwords, dtype=np.int64)
        live = np.nonzero((cnt > 0) & (key[seg_of_set[row_seg]] > 0))[0]
        rk = key[seg_of_set[row_seg[live]]]
        lost = np.zeros(len(set_ids), dtype=bool)
        words_of = []
        for r, k in zip(live, rk):
            w0, w1 = gs[r] >> 6, (ge[r] - 1) >> 6
            ws = [w for w in range(w0, w1 + 1) if unc[max(gs[r], w * 64):min(ge[r], w * 64 + 64)].any()]
            words_of.append(ws)
            for w in ws:
                if k > owner[w]: owner[w] = k
        for r, k, ws in zip(live, rk, words_of):
            if any(owner[w] != k for w in ws): lost[seg_of_set[row_seg[r]]] = True
        winners = np.nonzero((key > 0) & ~lost)[0]
        # per universe: weights of words by owner, T_u and M_u
        bounds = [cmax] if not LEVELS else [min(b, cmax) for b in (8, 16, 32, 64, 128)] + [cmax]
        T = np.full((U, len(bounds)), np.iinfo(np.int64).max); M = np.zeros(U, dtype=np.int64)
        for u in range(U):
            if need[u] <= 0 or goff[u + 1] == goff[u]: continue
            w0, w1 = goff[u] >> 6, (goff[u + 1] - 1) >> 6
            ks, wts = [], []
            for w in range(w0, w1 + 1):
                a, b = max(goff[u], w * 64), min(goff[u + 1], w * 64 + 64)
                wt = int(unc[a:b].sum())
                if wt: ks.append(owner[w]); wts.append(wt)
            if not ks: continue
            ks = np.array(ks); wts = np.array(wts)
            M[u] = ks.max()
            o = np.argsort(-ks, kind="stable")
            cum = np.cumsum(wts[o])
            for li, b in enumerate(bounds):
                x = need[u] - b
                if x >= 0:
                    j = int(np.searchsorted(cum, x, side="right"))   # first j with cum[j] > x
                    T[u, li] = 0 if j >= len(o) else ks[o[j]]
        accepted = []
        for c in winners:
            k = key[c]
            us = seg_u[setstart[c]:(setstart[c + 1] if c + 1 < len(setstart) else len(segstart))]
            sc = segcnt[setstart[c]:(setstart[c + 1] if c + 1 < len(setstart) else len(segstart))]
            ok = all((sc[i] == 0) or (k >= T[u, level_of(sc[i], bounds)]) or (k >= M[u]) for i, u in enumerate(us))
            if ok: accepted.append(c)
        assert accepted, "no progress"
        accepted.sort(key=lambda c: -key[c])
        for c in accepted:
            picks.append((int(key[c]), int(set_ids[c])))
            alive_set[c] = False
            for r in range(segstart[setstart[c]], segstart[setstart[c + 1]] if c + 1 < len(setstart) else len(rs)):
                n = int(unc[gs[r]:ge[r]].sum())
                unc[gs[r]:ge[r]] = False
                need[ru[r]] -= n
        for u in np.nonzero(need <= 0)[0]:
            unc[goff[u]:goff[u + 1]] = False
        per_round.append(len(accepted))
    # the sequential order: by accept-time key, descending (gains never grow; ties to the smaller id)
    picks.sort(key=lambda t: -t[0])
    return [t[1] for t in picks], rounds, per_round


if __name__ == "__main__":
    orc.build(); orc.set_threads(8)
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
    gsel = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 7, 3]
    groups = synthetic.dataset("S4", scale=scale)
    for gi in gsel:
        genomes = groups[gi]
        seqs = [s for g in genomes for s in g]
        cands = list(dict.fromkeys(candidate_probes.candidate_strings_from_sequences(seqs, 100, 50)))
        k, entries = orc.anchor_table(cands, 2, 100)
        rs, ru, st, en = orc.make_sets(cands, entries, k, genomes, 2, 100, 0, 50)
        glen = np.array([sum(len(s) for s in g) for g in genomes], dtype=np.int64)
        goff = np.concatenate(([0], np.cumsum(glen)))
        for p in (0.9, 0.5):
            ref = orc.lazy_greedy(rs, ru, st, en, len(cands), glen, universe_p=[p] * len(genomes))
            t0 = time.time()
            picks, rounds, per = simulate(rs, ru, st + goff[ru], en + goff[ru], goff, len(cands), p)
            print("group %d p=%.1f: %d sets %d rows %d universes: picks %d (oracle %d) %s; rounds %d; accepted per round: first %s last %s (%.0f s)"
                  % (gi, p, len(cands), len(rs), len(genomes), len(picks), len(ref),
                     "SAME ORDER" if picks == ref else ("same set" if sorted(picks) == sorted(ref) else "DIFFERENT"),
                     rounds, per[:8], per[-8:], time.time() - t0), flush=True)
