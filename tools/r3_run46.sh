#!/bin/bash
cd "$(dirname "$0")/.."
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "minhash or ndf or chain or config5 or config3 or union or design_large or lazy_resolution" > gpurun_out/run46_tests.txt 2>&1
tail -4 gpurun_out/run46_tests.txt
timeout 100 python bench.py --workload S3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('S3', b['ms_per_step'], b['roofline_k3']['device_ms_per_step'], b['roofline_k3']['pairs_compared'], b['parity_vs_golden_digests'])"
CATCHHIP_TIMING=2 timeout 200 python tools/s5_profile.py 1.0 once > gpurun_out/s5_profile_x1f.txt 2>&1
grep "minhash filter.*rounds\|^cluster" gpurun_out/s5_profile_x1f.txt | head -16 | cut -c1-150
grep -c "lazy round" gpurun_out/s5_profile_x1f.txt
grep "lazy round" gpurun_out/s5_profile_x1f.txt | head -8 | cut -c1-180
grep "ndf_minhash_many" gpurun_out/s5_profile_x1f.txt | head -2 | cut -c1-150
