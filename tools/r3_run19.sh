#!/bin/bash
# -c 0.9, group 0 of S4: per-round kernel durations (second repetition)
cd "$(dirname "$0")/.."
python -c "import __graft_entry__" 
out=$PWD/gpurun_out/c09prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python /root/repo/tools/c09_bench.py 0.9 0 > $out/c09_g0.txt 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/tr/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    for k in ('gr_count', 'gr_claim', 'gr_apply', 'gr_usel', 'gr_finish', 'gr_fixup'):
        if n.startswith(k) or (' ' + k) in n:
            per[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
nr = len(per['gr_claim']) // 2
print('rounds per rep', nr)
print('round count claim usel apply finish fixup (us)')
for i in list(range(0, 20)) + list(range(20, nr, 8)):
    j = nr + i
    print(i, *[round(per[k][len(per[k]) // 2 + i], 1) if len(per[k]) // 2 + i < len(per[k]) else None for k in ('gr_count', 'gr_claim', 'gr_usel', 'gr_apply', 'gr_finish', 'gr_fixup')])
for k in per:
    h = per[k][len(per[k]) // 2:]
    print(k, 'total ms', round(sum(h) / 1e3, 2), 'first16', round(sum(h[:16]) / 1e3, 2), 'rest', round(sum(h[16:]) / 1e3, 2))
PY
tail -2 $out/c09_g0.txt
