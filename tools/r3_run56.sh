#!/bin/bash
cd "$(dirname "$0")/.."
for sc in 0.25 0.5 1.0; do
  timeout 600 python bench.py --workload S5 --scale $sc --steps 1 --warmup 0 > gpurun_out/run56_s5_$sc.json 2> gpurun_out/run56_s5_$sc.err
  python - $sc <<'PY'
import json, sys
b=json.load(open('gpurun_out/run56_s5_%s.json' % sys.argv[1]))
print(sys.argv[1], round(b['ms_per_step']), {k: round(v, 2) for k, v in b.get('wall_s_per_step', {}).items()}, b['work_per_step']['probes'], b['probes_sha256'][:12], b.get('solver_families_agree'))
PY
done
PYTHONHASHSEED=0 timeout 1000 python tests/fuzz_parity.py 900 31337 > gpurun_out/run56_fuzz.txt 2>&1
tail -1 gpurun_out/run56_fuzz.txt | cut -c1-160
