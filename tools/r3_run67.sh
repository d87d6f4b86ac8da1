#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python bench.py --workload S4i > gpurun_out/run67_s4i.json 2> gpurun_out/run67_s4i.err
python -c "
import json; b=json.load(open('gpurun_out/run67_s4i.json')); print(b['ms_per_step'], b['kernel_ms_per_step'], b['parity_vs_golden_digests'], b['m2_setcoverfilter_wall_s'], b['partial_coverage']['ms_per_step'], b['partial_coverage']['parity_vs_golden_digests'], b['config']['one_instance'])"
