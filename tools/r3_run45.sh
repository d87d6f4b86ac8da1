#!/bin/bash
cd "$(dirname "$0")/.."
PYTHONHASHSEED=0 timeout 900 python tests/fuzz_parity.py 420 20260928 > gpurun_out/run45_fuzz.txt 2>&1
tail -6 gpurun_out/run45_fuzz.txt
