#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/pytest9.log
tail -5 gpurun_out/r3/pytest9.log
timeout 900 python bench.py --workload S5 --scale 0.25 --steps 1 --warmup 1 > gpurun_out/r3/s5_025v.json 2> gpurun_out/r3/s5_025v.err; echo "s5 rc=$?"
timeout 300 python bench.py --workload S3 --steps 5 --warmup 2 > gpurun_out/r3/s3b.json 2> gpurun_out/r3/s3b.err; echo "s3 rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/s5_025v.json').read().strip().splitlines()[-1])
for k in ("ms_per_step","wall_s_per_step","kernel_ms_per_step","solver_families_agree","probes_sha256","device_memory"): print(k, d.get(k))
d=json.loads(open('gpurun_out/r3/s3b.json').read().strip().splitlines()[-1])
for k in ("ms_per_step","parity_vs_golden_digests","parity_vs_oracle","roofline_k3","speedup_vs_cpu_oracle"): print(k, d.get(k))
PY
