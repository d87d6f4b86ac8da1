#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python bench.py > gpurun_out/run61_bench.json 2> gpurun_out/run61_bench.err
tail -3 gpurun_out/run61_bench.err
python -c "
import json; b=json.load(open('gpurun_out/run61_bench.json')); print(b['ms_per_step'], b['kernel_ms_per_step'], b['roofline']['kernel'][:20], round(b['roofline']['frac'],3), b['roofline']['launches_per_step'], b['m2_setcoverfilter_wall_s'], b['parity_vs_golden_digests'], b['parity_vs_oracle'], round(b['speedup_vs_cpu_oracle']), b['partial_coverage']['parity_vs_golden_digests'], b['groups_overlapped']['ms_per_step'], b['groups_overlapped']['parity_vs_golden_digests'], b['config']['one_instance'])"
CATCHHIP_BENCH_UNION_BELOW_MBASES=0 timeout 300 python bench.py --no-cpu-baseline --no-m2 --no-partial --no-overlap-figure 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('no union', b['ms_per_step'], b['kernel_ms_per_step'])"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 2>/dev/null | grep "^{" | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('2 ranks', b['ms_per_step'], b['parity_vs_golden_digests'])"
