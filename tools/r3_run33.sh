#!/bin/bash
cd "$(dirname "$0")/.."
for sh in 23 24 25; do
  echo "== CATCHHIP_FLAT_MAX_TILE_SHIFT=$sh"
  CATCHHIP_FLAT_MAX_TILE_SHIFT=$sh timeout 600 python bench.py --workload S4 --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['kernel_ms_per_step'], b['parity_vs_golden_digests'])"
done
echo "== partial default"
timeout 600 python tools/c09_bench.py 0.9 2>&1 | grep "^group  0\|coverage"
