#!/bin/bash
cd "$(dirname "$0")/.."
timeout 300 python tools/step_overhead2.py 2>&1 | tail -30 | cut -c1-160
