#!/bin/bash
cd "$(dirname "$0")/.."
for lim in 16 32 50; do timeout 300 python tools/union_small_groups.py $lim 2>&1 | tail -4; done
