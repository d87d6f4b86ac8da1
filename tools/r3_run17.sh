#!/bin/bash
# -c 0.9 on S4: kernel summary of the partial-coverage rounds (largest group and all groups)
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/c09prof; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python /root/repo/tools/c09_bench.py 0.9 > $out/c09.txt 2>&1
f=$(ls -t $out/trace/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv
head -30 $out/kernel_stats.csv | cut -c1-60,150-400
tail -25 $out/c09.txt
rm -rf $out/trace
