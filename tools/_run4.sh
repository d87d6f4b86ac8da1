export CATCHHIP_TEST_HOOKS=1
python -m pytest tests -m gpu -x -q -k "full_size_config5" 2>&1 | tail -5
python bench.py --workload S5 --scale 0.05 --steps 1 --warmup 1 --no-solver-check --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['ms_per_step'], d['parity_vs_golden_digests'], d['work_per_step']['probes'], d['wall_s_per_step'])"
