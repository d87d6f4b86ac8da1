#!/bin/bash
cd "$(dirname "$0")/.."
CATCHHIP_TIMING=1 timeout 300 python bench.py --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure --steps 2 --warmup 2 2>&1 >/dev/null | grep "filter:\|to queue\|solver batch" | tail -16
