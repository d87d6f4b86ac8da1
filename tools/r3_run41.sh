#!/bin/bash
# final evidence of the round: S4 (default bench + rocprofv3 + PMC) and S3
cd "$(dirname "$0")/.."
bash tools/collect_profiles.sh r03 S4 > gpurun_out/collect_r03_S4.txt 2>&1
bash tools/collect_profiles.sh r03s3 S3 > gpurun_out/collect_r03_S3.txt 2>&1
python - <<'PY'
import json
for d in ('profiles_r03','profiles_r03s3'):
    b=json.load(open('gpurun_out/%s/bench.json'%d))
    print(d, round(b['ms_per_step'],1), b['roofline']['kernel'][:30], round(b['roofline']['frac'],3), b.get('m2_setcoverfilter_wall_s'), b.get('speedup_vs_cpu_oracle'), b['parity_vs_golden_digests'], (b.get('partial_coverage') or {}).get('ms_per_step'))
PY
