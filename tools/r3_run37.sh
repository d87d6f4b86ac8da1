#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rccl or ranks or preflight or shard" 2>&1 | tail -4
