#!/bin/bash
cd "$(dirname "$0")/.."
for n in 2 4 8; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510+n)) bench.py --gpus $n --steps 2 --warmup 1 > gpurun_out/run43_n$n.json 2> gpurun_out/run43_n$n.err
  python - $n <<'PY'
import json, sys
n=sys.argv[1]
try:
    lines=[l for l in open('gpurun_out/run43_n%s.json'%n).read().splitlines() if l.startswith('{')]
    b=json.loads(lines[-1])
    print(n, 'ranks:', round(b['ms_per_step'],1), 'ms/step value', '%.3g'%b['value'], b['scaling'], b['parity_vs_golden_digests'], b['config'].get('sharded_groups'), b.get('rccl'))
except Exception as e:
    print(n, 'failed', e); print(open('gpurun_out/run43_n%s.err'%n).read()[-1500:])
PY
done
