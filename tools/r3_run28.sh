#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "minhash or ndf or chain or config5 or config3 or union or design_large or lazy_resolution" > gpurun_out/run28_tests.txt 2>&1
tail -5 gpurun_out/run28_tests.txt
timeout 600 python bench.py --workload S3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['roofline_k3']['device_ms_per_step'], b['roofline_k3']['pairs_compared'], b['parity_vs_golden_digests'])"
CATCHHIP_TIMING=2 timeout 1200 python tools/s5_profile.py 1.0 once > gpurun_out/s5_profile_x1c.txt 2>&1
grep "minhash filter.*rounds\|^cluster" gpurun_out/s5_profile_x1c.txt | head -30 | cut -c1-170
grep "lazy round" gpurun_out/s5_profile_x1c.txt | head -48 | tail -12 | cut -c1-170
grep "ndf_minhash_many\|setcover_filter_many\|neighbors_many" gpurun_out/s5_profile_x1c.txt | head -5 | cut -c1-150
