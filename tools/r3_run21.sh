#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "minhash or ndf or chain or config5 or union or design_large or cluster" > gpurun_out/run21_tests.txt 2>&1
tail -5 gpurun_out/run21_tests.txt
timeout 900 python bench.py --workload S5 --scale 0.25 --steps 1 --warmup 0 > gpurun_out/run21_s5_025.json 2> gpurun_out/run21_s5_025.err
python - <<'PY'
import json
b=json.load(open('gpurun_out/run21_s5_025.json'))
print(b['ms_per_step'], b.get('wall_s_per_step'), b['work_per_step']['probes'], b.get('probes_sha256'))
PY
