#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/run34_tests.txt 2>&1
tail -4 gpurun_out/run34_tests.txt
timeout 600 python bench.py --workload S4 --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['kernel_ms_per_step'], b['parity_vs_golden_digests'])"
