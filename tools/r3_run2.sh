#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/pytest2.log
tail -4 gpurun_out/r3/pytest2.log
bash tools/prof_variants.sh lds ""
bash tools/prof_variants.sh global "CATCHHIP_FLAT_CHG_GLOBAL=1"
bash tools/prof_variants.sh nocache "CATCHHIP_FLAT_NOCACHE=1"
bash tools/prof_variants.sh s3 "" --workload S3
timeout 900 python bench.py --workload S5 --scale 0.25 --steps 1 --warmup 1 > gpurun_out/r3/s5_025u.json 2> gpurun_out/r3/s5_025u.err; echo "s5 rc=$?"; tail -c 300 gpurun_out/r3/s5_025u.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/s5_025u.json').read().strip().splitlines()[-1])
for k in ("ms_per_step","wall_s_per_step","kernel_ms_per_step","work_per_step","solver_families_agree","other_family_wall_s","probes_sha256","device_memory","config"): print(k, d.get(k))
PY
