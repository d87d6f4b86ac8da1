#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for sc in 0.25 1.0; do
  timeout 600 python bench.py --workload S5 --scale $sc --steps 1 --warmup 0 > gpurun_out/run47_s5_$sc.json 2> gpurun_out/run47_s5_$sc.err
  python - $sc <<'PY'
import json, sys
b=json.load(open('gpurun_out/run47_s5_%s.json' % sys.argv[1]))
print(sys.argv[1], round(b['ms_per_step']), {k: round(v, 2) for k, v in b.get('wall_s_per_step', {}).items()}, b['work_per_step']['probes'], b['probes_sha256'][:12], b.get('solver_families_agree'))
PY
done
