#!/bin/bash
cd "$(dirname "$0")/.."
CATCHHIP_TIMING=2 timeout 600 python bench.py --workload S3 --no-cpu-baseline --steps 1 --warmup 1 2>&1 | grep "lazy round" | tail -45 | cut -c1-170
