#!/bin/bash
cd "$(dirname "$0")/.."
CATCHHIP_TIMING=2 timeout 1200 python tools/s5_profile.py 1.0 once > gpurun_out/s5_profile_x1b.txt 2>&1
grep "minhash filter\|lazy round" gpurun_out/s5_profile_x1b.txt | head -45 | cut -c1-170
