#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --workload S4 --no-cpu-baseline --no-partial --no-overlap-figure 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print(round(b['ms_per_step'],1), 'M2', b['m2_setcoverfilter_wall_s'], b['m2_serial_wall_s'], b['m2_parity_vs_golden_digests'], b['parity_vs_golden_digests'])"
CATCHHIP_UNION_SMALL_BELOW_MBASES=0 timeout 300 python bench.py --workload S4 --no-cpu-baseline --no-partial --no-overlap-figure 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('no union: M2', b['m2_setcoverfilter_wall_s'], b['m2_serial_wall_s'], b['m2_parity_vs_golden_digests'])"
