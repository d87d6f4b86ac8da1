#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/pytest5.log
tail -25 gpurun_out/r3/pytest5.log
timeout 600 python tools/c09_bench.py 0.9 > gpurun_out/r3/c09c.log 2>&1; tail -22 gpurun_out/r3/c09c.log
