#!/bin/bash
# set-iteration order on the device: test, then the S3 bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "set_iteration_order or ndf_then_scf or full_size_config3 or ndf" > gpurun_out/run16_tests.txt 2>&1
tail -5 gpurun_out/run16_tests.txt
timeout 600 python bench.py --workload S3 > gpurun_out/run16_s3.json 2> gpurun_out/run16_s3.err
python - <<'PY'
import json
b=json.load(open('gpurun_out/run16_s3.json'))
print(b['ms_per_step'], b['roofline'], b['parity_vs_golden_digests'], b.get('speedup_vs_cpu_oracle'))
PY
CATCHHIP_PYSET_DEVICE_FROM=4000000000 timeout 600 python bench.py --workload S3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('host order:', b['ms_per_step'])"
