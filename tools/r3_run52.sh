#!/bin/bash
cd "$(dirname "$0")/.."
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "partial or full_size_config4 or universe_p or coverage or greedy" 2>&1 | tail -3
timeout 200 python tools/c09_bench.py 0.9 2>&1 | grep "^group  0\|coverage"
