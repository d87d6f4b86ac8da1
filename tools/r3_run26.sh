#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "minhash or ndf or chain or config5 or config3 or union or design_large or cluster or lazy_resolution or neighbor" > gpurun_out/run26_tests.txt 2>&1
tail -15 gpurun_out/run26_tests.txt
timeout 600 python bench.py --workload S3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['roofline_k3'], b['parity_vs_golden_digests'])"
