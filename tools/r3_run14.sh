#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q -k "prefetched or live_reference or preflight or full_size_config4" > gpurun_out/r3/pytest14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/pytest14.log
tail -4 gpurun_out/r3/pytest14.log
for w in 4 3 2; do
CATCHHIP_GROUPS_IN_FLIGHT=$w timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-partial --no-overlap-figure --m2-steps 6 > gpurun_out/r3/m2s_w$w.json 2> gpurun_out/r3/m2s_w$w.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3/m2s_w$w.json').read().strip().splitlines()[-1])
print("static lanes width=$w", d["m2_setcoverfilter_wall_s"], [round(x,4) for x in d["m2_steps_s"]], d["m2_serial_wall_s"], d["m2_parity_vs_golden_digests"], d["device_memory"])
PY
done
