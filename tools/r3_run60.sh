#!/bin/bash
cd "$(dirname "$0")/.."
for t in 0 8 16 24; do
CATCHHIP_UNION_SMALL_BELOW_MBASES=$t timeout 300 python bench.py --workload S4 --no-cpu-baseline --no-partial --no-overlap-figure --m2-steps 5 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('union below $t Mbases: M2', round(b['m2_setcoverfilter_wall_s'],4), 'serial', round(b['m2_serial_wall_s'],4), b['m2_parity_vs_golden_digests'])"
done
