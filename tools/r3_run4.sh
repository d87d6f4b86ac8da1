#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/pytest4.log
tail -15 gpurun_out/r3/pytest4.log
timeout 900 python tools/c09_bench.py 0.9 > gpurun_out/r3/c09b.log 2>&1; tail -22 gpurun_out/r3/c09b.log
