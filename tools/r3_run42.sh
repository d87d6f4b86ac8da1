#!/bin/bash
cd "$(dirname "$0")/.."
t0=$(date +%s)
python bench.py > gpurun_out/run42_bench.json 2> gpurun_out/run42_bench.err
t1=$(date +%s); echo "default bench wall: $((t1-t0)) s"
python -c "
import json; b=json.load(open('gpurun_out/run42_bench.json')); print(b['ms_per_step'], b['kernel_ms_per_step'], b['m2_setcoverfilter_wall_s'], b['parity_vs_golden_digests'], b['parity_vs_oracle'], b['speedup_vs_cpu_oracle'])"
