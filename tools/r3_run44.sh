#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python tools/shard_loop_time.py 0 2>&1 | tail -4
timeout 600 python tools/shard_loop_time.py 7 2>&1 | tail -2
