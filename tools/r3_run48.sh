#!/bin/bash
cd "$(dirname "$0")/.."
CATCHHIP_FLAT_TRACE=1 timeout 60 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_greedy_batched_rounds_restore_sequential_order and True" > gpurun_out/run48.txt 2>&1
head -40 gpurun_out/run48.txt | cut -c1-220
