#!/bin/bash
# round-3 GPU call 1: tests, pack profile, S4 bench (M2 pipelined), S3 with NDF, S5 scaled
mkdir -p gpurun_out/r3
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/pytest1.log
tail -5 gpurun_out/r3/pytest1.log
CATCHHIP_TIMING=1 timeout 300 python tools/pack_profile.py > gpurun_out/r3/pack1.log 2> gpurun_out/r3/pack1.err; tail -3 gpurun_out/r3/pack1.log
grep "gathered" gpurun_out/r3/pack1.err | sort -t' ' -k4 -n | tail -3
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r3/b1.json 2> gpurun_out/r3/b1.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r3/b1.err
timeout 300 python bench.py --workload S3 --steps 5 --warmup 2 > gpurun_out/r3/s3.json 2> gpurun_out/r3/s3.err; echo "s3 rc=$?"; tail -c 600 gpurun_out/r3/s3.err
timeout 300 python bench.py --workload S5 --scale 0.05 --steps 2 --warmup 1 > gpurun_out/r3/s5_005.json 2> gpurun_out/r3/s5_005.err; echo "s5 rc=$?"; tail -c 600 gpurun_out/r3/s5_005.err
timeout 600 python bench.py --workload S5 --scale 0.25 --steps 1 --warmup 1 > gpurun_out/r3/s5_025.json 2> gpurun_out/r3/s5_025.err; echo "s5b rc=$?"; tail -c 600 gpurun_out/r3/s5_025.err
