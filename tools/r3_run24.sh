#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cluster or config5 or design_large or neighbour or neighbor" 2>&1 | tail -3
timeout 900 python tools/s5_profile.py 0.25 once 2>&1 | grep -v "^\[catchhip\]\|^MinHash" | head -30 | cut -c1-170
