#!/bin/bash
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/run51_bench.json 2> gpurun_out/run51_bench.err
python -c "
import json; b=json.load(open('gpurun_out/run51_bench.json')); print(b['ms_per_step'], b['roofline']['kernel'][:24], round(b['roofline']['frac'],3), b['m2_setcoverfilter_wall_s'], b['parity_vs_golden_digests'], b['parity_vs_oracle'], round(b['speedup_vs_cpu_oracle']))"
