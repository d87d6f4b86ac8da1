#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cluster or config5 or design_large or neighbor" 2>&1 | tail -3
timeout 1200 python tools/s5_profile.py 1.0 once > gpurun_out/s5_profile_x1e.txt 2>&1
grep "^cluster\|components search" gpurun_out/s5_profile_x1e.txt
grep "ndf_minhash_many\|setcover_filter_many\|neighbors_many\|_components\|signatures\|engine.py:681" gpurun_out/s5_profile_x1e.txt | head -12 | cut -c1-150
