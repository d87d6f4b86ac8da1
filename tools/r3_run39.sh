#!/bin/bash
cd "$(dirname "$0")/.."
for v in base c5 c6 k6 k8; do
  if [ $v = base ]; then unset CATCHHIP_LIB; else export CATCHHIP_LIB=$PWD/gpurun_variants/libcatchhip_$v.so; fi
  timeout 600 python bench.py --workload S4 --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$v', round(b['ms_per_step'],1), round(b['kernel_ms_per_step']['k2_greedy_rounds_only'],1), 'claim', round(b['roofline']['device_ms_per_step'],2), b['parity_vs_golden_digests'])"
done
