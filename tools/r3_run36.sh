#!/bin/bash
cd "$(dirname "$0")/.."
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/run36_tests.txt 2>&1
tail -3 gpurun_out/run36_tests.txt
for v in 0 1; do
  if [ $v = 1 ]; then export CATCHHIP_SEED_NO_PRESENCE=1; fi
  echo "== CATCHHIP_SEED_NO_PRESENCE=$v"
  timeout 600 python bench.py --workload S4 --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['kernel_ms_per_step']['k1_scan'], b['kernel_ms_per_step']['k1_seed_verify'], b['parity_vs_golden_digests'])"
done
