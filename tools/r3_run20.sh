#!/bin/bash
cd "$(dirname "$0")/.."
CATCHHIP_TIMING=1 timeout 1200 python tools/s5_profile.py 1.0 once > gpurun_out/s5_profile_x1.txt 2>&1
grep -v "^\[catchhip\] gather" gpurun_out/s5_profile_x1.txt | head -90
