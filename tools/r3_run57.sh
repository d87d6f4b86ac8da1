#!/bin/bash
cd "$(dirname "$0")/.."
for mr in 4194304 2000000 1000000 300000 100000; do
  CATCHHIP_FLAT_MIN_ROWS=$mr timeout 300 python bench.py --workload S4 --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$mr', round(b['ms_per_step'],1), round(b['kernel_ms_per_step']['k2_greedy'],1), round(b['kernel_ms_per_step']['k2_greedy_rounds_only'],1), b['parity_vs_golden_digests'])"
done
