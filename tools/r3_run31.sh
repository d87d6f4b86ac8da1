#!/bin/bash
# full GPU suite, then the bench lines of the round (S4 default, S3, S5 at three scales)
cd "$(dirname "$0")/.."
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/run31_tests.txt 2>&1
tail -4 gpurun_out/run31_tests.txt
for sc in 0.25 0.5 1.0; do
  timeout 1500 python bench.py --workload S5 --scale $sc --steps 1 --warmup 0 > gpurun_out/run31_s5_$sc.json 2> gpurun_out/run31_s5_$sc.err
  python - $sc <<'PY'
import json, sys
b=json.load(open('gpurun_out/run31_s5_%s.json' % sys.argv[1]))
print(sys.argv[1], round(b['ms_per_step']), {k: round(v, 2) for k, v in b.get('wall_s_per_step', {}).items()}, b['work_per_step']['probes'], b['probes_sha256'][:12], b.get('solver_families_agree'))
PY
done
timeout 900 python bench.py --workload S3 > gpurun_out/run31_s3.json 2> gpurun_out/run31_s3.err
python -c "
import json; b=json.load(open('gpurun_out/run31_s3.json')); print('S3', b['ms_per_step'], b['roofline'], b['parity_vs_golden_digests'], b.get('speedup_vs_cpu_oracle'))"
