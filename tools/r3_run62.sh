#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python tools/union_small_groups.py 1000 2>&1 | tail -4
