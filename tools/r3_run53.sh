#!/bin/bash
cd "$(dirname "$0")/.."
CATCHHIP_FLAT_MIN_ROWS=0 CATCHHIP_PARTIAL_MIN_ROWS=0 PYTHONHASHSEED=0 timeout 400 python tests/fuzz_parity.py 300 777000 > gpurun_out/run53_fuzz.txt 2>&1
tail -2 gpurun_out/run53_fuzz.txt | cut -c1-200
