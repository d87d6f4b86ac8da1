#!/bin/bash
cd "$(dirname "$0")/.."
timeout 300 python bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('steps1', b['ms_per_step'], b['steps'], b['warmup'], b['parity_vs_golden_digests'])"
timeout 300 python bench.py --workload S2 --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('S2', b['ms_per_step'], b['parity_vs_oracle'], b.get('speedup_vs_cpu_oracle'))"
timeout 300 python bench.py --workload S1 --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('S1', b['ms_per_step'], b['parity_vs_oracle'])"
