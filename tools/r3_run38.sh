#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_config4" 2>&1 | tail -3
timeout 900 python bench.py --workload S4i > gpurun_out/run38_s4i.json 2> gpurun_out/run38_s4i.err
python -c "
import json; b=json.load(open('gpurun_out/run38_s4i.json')); print(b['ms_per_step'], b['kernel_ms_per_step'], b['parity_vs_golden_digests'], b.get('m2_setcoverfilter_wall_s'), b.get('partial_coverage'), b.get('speedup_vs_cpu_oracle'))"
