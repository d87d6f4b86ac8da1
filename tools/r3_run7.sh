#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/pytest7.log
tail -5 gpurun_out/r3/pytest7.log
timeout 600 python tools/s5_profile.py 0.25 > gpurun_out/r3/s5prof2.log 2>&1; head -40 gpurun_out/r3/s5prof2.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r3/b2.json 2> gpurun_out/r3/b2.err; echo "bench rc=$?"; tail -c 300 gpurun_out/r3/b2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/b2.json').read().strip().splitlines()[-1])
for k in ("value","ms_per_step","m2_setcoverfilter_wall_s","m2_serial_wall_s","m2_parity_vs_golden_digests","parity_vs_golden_digests","parity_vs_oracle","gpu_ms_on_cpu_sample","speedup_vs_cpu_oracle","kernel_ms_per_step","partial_coverage","roofline"): print(k, d.get(k))
PY
