#!/bin/bash
cd "$(dirname "$0")/.."
CATCHHIP_TIMING=2 timeout 1200 python tools/s5_profile.py 1.0 once > gpurun_out/s5_profile_x1d.txt 2>&1
grep "lazy round\|minhash filter" gpurun_out/s5_profile_x1d.txt | head -50 | cut -c1-190
