#!/bin/bash
# one S4 bench line per environment setting given as arguments ("" = none): tools/quick_s4.sh "" "CATCHHIP_X=1"
for e in "$@"; do
env $e python bench.py --workload S4 --steps 3 --warmup 1 --no-also --no-property-checks 2>/dev/null | E="$e" python -c "
import sys, json, os
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print(os.environ['E'] or '-', d['ms_per_step'], {k: round(v, 2) for k, v in d.get('kernel_ms_per_step', {}).items() if isinstance(v, (int, float))})
"
done
