"""Seeded synthetic viral-like genome sets (new code; shapes from SURVEY.md §8(d)).

The reference ships no inputs at scale (its probe-designs/ are LFS stubs and
its datasets need network), so benchmarks and parity tests run on genomes
produced here: a random ACGT root per species, a 2-level tree (clades at
divergence d1 from the root, strains at d2 from their clade) by independent
point substitutions, plus N runs (per genome Poisson(0.5) runs of length
U[1,200] and Poisson(2) isolated N).  Upper-case only, alphabet {A,C,G,T,N} --
what catch/utils/seq_io.py:104-175 would hand to the filters.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mutate(codes, d, rng):
    if d <= 0:
        return codes.copy()
    mask = rng.random(codes.size) < d
    shift = rng.integers(1, 4, size=codes.size, dtype=np.uint8)
    out = codes.copy()
    out[mask] = (codes[mask] + shift[mask]) & 3
    return out


def _indels(codes, rate, rng):
    """Insertions and deletions: Poisson(rate * length) events, each at a
    uniform position, deletion or insertion with equal probability, length
    geometric (mean ~3, at most 30); applied from the end so that positions
    stay valid.  Sibling strains then no longer sit at identical offsets."""
    n = int(rng.poisson(rate * codes.size))
    if n == 0:
        return codes
    pos = np.sort(rng.integers(0, codes.size, size=n))[::-1]
    kinds = rng.random(n) < 0.5
    lens = np.minimum(rng.geometric(0.35, size=n), 30)
    out = codes
    for p, is_del, ln in zip(pos.tolist(), kinds.tolist(), lens.tolist()):
        if is_del:
            out = np.concatenate((out[:p], out[p + ln:]))
        else:
            out = np.concatenate((out[:p], rng.integers(0, 4, size=ln, dtype=np.uint8), out[p:]))
    return out


def _to_str(codes, rng, with_n=True):
    b = _ACGT[codes].copy()
    if with_n:
        for _ in range(int(rng.poisson(0.5))):
            s = int(rng.integers(0, codes.size))
            ln = int(rng.integers(1, 201))
            b[s:s + ln] = ord("N")
        for _ in range(int(rng.poisson(2))):
            b[int(rng.integers(0, codes.size))] = ord("N")
    return b.tobytes().decode("ascii")


def make_species(rng, segment_lengths, n_strains, n_clades, d1, d2,
                 with_n=True, indel=0.0):
    """Returns a list of genomes; each genome is a list of segment strings.
    indel > 0: insertions / deletions per base at the clade and at the strain
    level (the S*i datasets; 0 leaves the random stream of S1-S5 untouched)."""
    roots = [rng.integers(0, 4, size=ln, dtype=np.uint8)
             for ln in segment_lengths]
    if indel <= 0:
        clades = [[_mutate(r, d1, rng) for r in roots] for _ in range(n_clades)]
    else:
        clades = [[_indels(_mutate(r, d1, rng), indel, rng) for r in roots] for _ in range(n_clades)]
    genomes = []
    for i in range(n_strains):
        c = clades[i % n_clades]
        if indel <= 0:
            genomes.append([_to_str(_mutate(seg, d2, rng), rng, with_n)
                            for seg in c])
        else:
            genomes.append([_to_str(_indels(_mutate(seg, d2, rng), indel, rng), rng, with_n)
                            for seg in c])
    return genomes


def dataset(name, seed=None, scale=1.0):
    """Named synthetic inputs mirroring BASELINE.json's configs.

    Returns a list of groups; each group is a list of genomes; each genome is
    a list of sequence strings (segments/chromosomes).

      S1: 1 species, 10,800 bp, 1 genome (config 1)
      S2: 2 species = 2 groups: 18,950 bp x 60 strains; (7,270 + 3,400) bp
          x 40 strains; d1 = 5 %, d2 = 1 % (config 2, ~1.56 Mbp)
      S3: 8 segments x (5,000*scale) strains in 40 clades, every segment its
          own genome record, d1 = 12 %, d2 = 2 % (config 3 shape)
      S4: 20 species = 20 groups (config 4 shape), sizes scaled by `scale`
      S2i, S4i: the shapes of S2 / S4 with per-clade and per-strain insertions
          and deletions (Poisson, 1 per kb; lengths geometric)
      S5: 588 species in ONE group (config 5 shape: `design_large` clusters
          all sequences itself), strain counts Zipf(1.3) capped at 20,000,
          scaled by `scale`; genome lengths log-uniform 3-200 kb; at scale 1
          about 2 x 10^9 bases
      S5m: the first 40 species of S5 (for timing the live reference's
          design_large chain)
    """
    if name == "S1":
        rng = np.random.Generator(np.random.PCG64(1 if seed is None else seed))
        return [make_species(rng, [10800], 1, 1, 0.0, 0.0)]
    if name == "S2":
        rng = np.random.Generator(np.random.PCG64(2 if seed is None else seed))
        n1 = max(1, int(round(60 * scale)))
        n2 = max(1, int(round(40 * scale)))
        return [make_species(rng, [18950], n1, 4, 0.05, 0.01),
                make_species(rng, [7270, 3400], n2, 4, 0.05, 0.01)]
    if name == "S2i":
        # S2's shape with insertions / deletions (1 per kb at both levels): hits of a probe in sibling
        # strains are no longer at identical offsets
        rng = np.random.Generator(np.random.PCG64(12 if seed is None else seed))
        n1 = max(1, int(round(60 * scale)))
        n2 = max(1, int(round(40 * scale)))
        return [make_species(rng, [18950], n1, 4, 0.05, 0.01, indel=1e-3),
                make_species(rng, [7270, 3400], n2, 4, 0.05, 0.01, indel=1e-3)]
    if name == "S4i":
        # S4's shape (20 species = 20 groups) with insertions / deletions
        rng = np.random.Generator(np.random.PCG64(14 if seed is None else seed))
        groups = []
        for i in range(20):
            ln = int(np.exp(rng.uniform(np.log(7000), np.log(30000))))
            if i == 0:
                ln = 150000
            n = max(1, int(round(int(rng.integers(50, 2001)) * scale)))
            groups.append(make_species(rng, [ln], n, min(8, n), 0.08, 0.015, indel=1e-3))
        return groups
    if name == "S3":
        rng = np.random.Generator(np.random.PCG64(3 if seed is None else seed))
        n = max(1, int(round(5000 * scale)))
        sp = make_species(rng, [2341, 2341, 2233, 1778, 1565, 1413, 1027, 890],
                          n, min(40, n), 0.12, 0.02)
        # every segment record is its own Genome (seq_io.py:97-101)
        return [[[seg] for g in sp for seg in g]]
    if name == "S4":
        rng = np.random.Generator(np.random.PCG64(4 if seed is None else seed))
        groups = []
        for i in range(20):
            ln = int(np.exp(rng.uniform(np.log(7000), np.log(30000))))
            if i == 0:
                ln = 150000
            n = max(1, int(round(int(rng.integers(50, 2001)) * scale)))
            groups.append(make_species(rng, [ln], n, min(8, n), 0.08, 0.015))
        return groups
    if name in ("S5", "S5m"):
        # S5m: the first 40 species of the same stream (a few Mbases: what the Python reference's
        # design_large chain finishes in minutes)
        rng = np.random.Generator(np.random.PCG64(5 if seed is None else seed))
        genomes = []
        for _ in range(588 if name == "S5" else 40):
            ln = int(np.exp(rng.uniform(np.log(3000), np.log(200000))))
            n = min(int(rng.zipf(1.3)), 20000)
            # keep the target of ~2e9 bases at scale 1: long genomes get fewer strains
            n = max(1, int(round(min(n, 4.0e7 / ln) * scale)))
            genomes += make_species(rng, [ln], n, min(6, n), 0.10, 0.02)
        return [genomes]
    raise ValueError("unknown synthetic dataset %r" % (name,))
