#!/bin/bash
cd "$(dirname "$0")/.."
for sh in 23 24 25; do
  echo "== CATCHHIP_FLAT_TILE_SHIFT=$sh"
  CATCHHIP_FLAT_TILE_SHIFT=$sh timeout 600 python tools/c09_bench.py 0.9 0,7 2>&1 | grep "^group\|coverage"
done
echo "== default"
timeout 600 python tools/c09_bench.py 0.9 0,7 2>&1 | grep "^group\|coverage"
