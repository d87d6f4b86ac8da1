"""Summary of CATCHHIP_TIMING=2 traces of the MinHash filter (stderr of tools/s5_profile.py): per file the
phase totals and, for the first chunks, the rounds.   python tools/ndf_trace_summary.py file.err ..."""
import re, sys
for fn in sys.argv[1:]:
    txt = open(fn).read().splitlines()
    chunks = []; cur = None
    tot = {'k-mers': 0, 'signatures + sorts': 0, 'rounds': 0}
    for l in txt:
        m = re.search(r"lazy round (\d+): (\d+) entries -> (\d+) undecided probes, (\d+) entries next; ([\d.]+) ms, (\d+) pairs", l)
        if m:
            r = int(m.group(1))
            if r == 0:
                cur = []; chunks.append(cur)
            cur.append((r, int(m.group(2)), float(m.group(5)), int(m.group(6))))
        m = re.search(r"tables: (k-mers|signatures \+ sorts|rounds) ([\d.]+) ms", l)
        if m:
            tot[m.group(1)] += float(m.group(2))
    print(fn, {k: round(v, 1) for k, v in tot.items()}, "chunks", len(chunks), "round-0 total %.1f" % sum(c[0][2] for c in chunks))
    for c in chunks[:3]:
        print("  %d rounds; r0 %.1f ms; rest %.1f ms; pairs %d" % (len(c), c[0][2], sum(x[2] for x in c[1:]), c[-1][3]))
        print("    ", [(x[1], x[2]) for x in c[1:70:4]])
