#!/bin/bash
cd "$(dirname "$0")/.."
CATCHHIP_FLAT_TRACE=2 timeout 600 python tools/c09_bench.py 0.9 0 2>&1 | grep -v "^W\|^E" | tail -75 | head -40
