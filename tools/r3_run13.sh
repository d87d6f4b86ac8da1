#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/pytest13.log
tail -4 gpurun_out/r3/pytest13.log
for lanes in 1 0; do
CATCHHIP_GROUP_LANES=$lanes timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-partial --no-overlap-figure --m2-steps 5 > gpurun_out/r3/m2_lanes$lanes.json 2> gpurun_out/r3/m2_lanes$lanes.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3/m2_lanes$lanes.json').read().strip().splitlines()[-1])
print("lanes=$lanes", d["ms_per_step"], d["m2_setcoverfilter_wall_s"], d["m2_steps_s"], d["m2_serial_wall_s"], d["m2_parity_vs_golden_digests"])
PY
done
for w in 2 3 6; do
CATCHHIP_GROUPS_IN_FLIGHT=$w timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-partial --no-overlap-figure --m2-steps 5 > gpurun_out/r3/m2_w$w.json 2> gpurun_out/r3/m2_w$w.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3/m2_w$w.json').read().strip().splitlines()[-1])
print("width=$w", d["m2_setcoverfilter_wall_s"], d["m2_steps_s"], d["m2_parity_vs_golden_digests"])
PY
done
for sc in 0.01 0.03; do
PYTHONHASHSEED=0 timeout 600 python bench.py --workload S5m --scale $sc --steps 2 --warmup 1 > gpurun_out/r3/s5m_$sc.json 2> gpurun_out/r3/s5m_$sc.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3/s5m_$sc.json').read().strip().splitlines()[-1])
print("S5m $sc", d["ms_per_step"], d["wall_s_per_step"], d["probes_sha256"], d["work_per_step"]["probes"], d["solver_families_agree"])
PY
done
