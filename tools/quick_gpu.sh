#!/bin/bash
# tests + the three bench workloads, one line each (GPU box; used between kernel changes)
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for wl in S4 S4i S3; do
python bench.py --workload $wl --steps 3 --warmup 1 --no-also --no-property-checks 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print('$wl', d['ms_per_step'], {k: round(v, 2) for k, v in d.get('kernel_ms_per_step', {}).items() if isinstance(v, (int, float))}, d.get('digests_equal', d.get('config', {}).get('digest')))
"
done
