#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/pytest3.log
tail -4 gpurun_out/r3/pytest3.log
bash tools/prof_variants.sh alive ""
timeout 600 python tools/c09_bench.py 0.9 > gpurun_out/r3/c09.log 2>&1; tail -22 gpurun_out/r3/c09.log
timeout 600 python tools/s5_profile.py 0.25 > gpurun_out/r3/s5prof.log 2>&1; head -80 gpurun_out/r3/s5prof.log
