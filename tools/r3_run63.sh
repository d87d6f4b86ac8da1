#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python bench.py > gpurun_out/run63_bench.json 2> gpurun_out/run63_bench.err
timeout 300 python bench.py --workload S3 > gpurun_out/run63_s3.json 2> gpurun_out/run63_s3.err
python -c "
import json
b=json.load(open('gpurun_out/run63_bench.json')); print(b['ms_per_step'], b['roofline'])
b=json.load(open('gpurun_out/run63_s3.json')); print(b['ms_per_step'], b['roofline'])"
