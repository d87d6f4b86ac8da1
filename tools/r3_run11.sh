#!/bin/bash
mkdir -p gpurun_out/r3
timeout 1200 python bench.py --workload S5 --scale 0.5 --steps 1 --warmup 0 --no-solver-check > gpurun_out/r3/s5_05.json 2> gpurun_out/r3/s5_05.err; echo "s5 0.5 rc=$?"; tail -c 400 gpurun_out/r3/s5_05.err
timeout 2400 python bench.py --workload S5 --scale 1.0 --steps 1 --warmup 0 --no-solver-check > gpurun_out/r3/s5_10.json 2> gpurun_out/r3/s5_10.err; echo "s5 1.0 rc=$?"; tail -c 600 gpurun_out/r3/s5_10.err
python - <<'PY'
import json
for f in ("s5_05","s5_10"):
    try:
        d=json.loads(open('gpurun_out/r3/%s.json'%f).read().strip().splitlines()[-1])
        for k in ("ms_per_step","wall_s_per_step","kernel_ms_per_step","work_per_step","device_memory","config"): print(f, k, d.get(k))
    except Exception as e: print(f, "ERR", e)
PY
