#!/bin/bash
cd "$(dirname "$0")/.."
timeout 400 python bench.py --no-cpu-baseline --no-m2 --no-overlap-figure 2> gpurun_out/run64.err | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['partial_coverage'])"
tail -3 gpurun_out/run64.err
