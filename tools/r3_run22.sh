#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "minhash or ndf or chain or config5 or union or design_large" > gpurun_out/run22_tests.txt 2>&1
tail -3 gpurun_out/run22_tests.txt
CATCHHIP_TIMING=1 timeout 900 python tools/s5_profile.py 0.25 once 2>&1 | grep "minhash\|to queue\|filter:\|^cluster" | head -60 | cut -c1-170
