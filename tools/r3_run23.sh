#!/bin/bash
cd "$(dirname "$0")/.."
CATCHHIP_TIMING=1 timeout 1500 python bench.py --workload S5 --scale 1.0 --steps 1 --warmup 0 > gpurun_out/run23_s5_1.json 2> gpurun_out/run23_s5_1.err
grep "minhash filter\|to queue" gpurun_out/run23_s5_1.err | awk '{ if ($0 ~ /rounds [0-9.]+ ms/ || $0 ~ /to queue/) print }' | cut -c1-150 | head -40
python - <<'PY'
import json
b=json.load(open('gpurun_out/run23_s5_1.json'))
print(b['ms_per_step'], b.get('wall_s_per_step'), b['work_per_step']['probes'], b.get('probes_sha256'))
PY
