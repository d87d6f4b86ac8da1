#!/bin/bash
bash tools/collect_profiles.sh r03 S4 > gpurun_out/collect_r03_S4.txt 2>&1
bash tools/collect_profiles.sh r03s3 S3 > gpurun_out/collect_r03_S3.txt 2>&1
tail -30 gpurun_out/collect_r03_S4.txt | cut -c1-300
