#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_config5" 2>&1 | tail -3
for sc in 0.25 0.5 1.0; do
  timeout 1500 python bench.py --workload S5 --scale $sc --steps 1 --warmup 0 > gpurun_out/run25_s5_$sc.json 2> gpurun_out/run25_s5_$sc.err
  python - $sc <<'PY'
import json, sys
b=json.load(open('gpurun_out/run25_s5_%s.json' % sys.argv[1]))
print(sys.argv[1], b['ms_per_step'], b.get('wall_s_per_step'), b['work_per_step']['probes'], b['config']['clusters'], {k: b.get(k) for k in ('property_checks','solver_families_agree')})
PY
done
timeout 600 python bench.py --workload S5 --scale 0.01 --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('0.01', b['ms_per_step'], b['parity_vs_golden_digests'])"
