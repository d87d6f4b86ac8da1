#!/bin/bash
cd "$(dirname "$0")/.."
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "minhash or ndf or chain or config5 or config3 or union or design_large or lazy_resolution" 2>&1 | tail -3
timeout 100 python bench.py --workload S3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('S3', b['ms_per_step'], b['roofline_k3']['device_ms_per_step'], b['parity_vs_golden_digests'])"
CATCHHIP_TIMING=1 timeout 200 python tools/s5_profile.py 1.0 once 2>&1 | grep "signatures + sorts\|^cluster\|ndf_minhash_many" | head -6 | cut -c1-150
