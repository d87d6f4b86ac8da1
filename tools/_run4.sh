export CATCHHIP_TEST_HOOKS=1
python -m pytest tests -m gpu -x -q -k "minhash or ndf or config5 or chains or candidates or front_end or design" 2>&1 | tail -3
CATCHHIP_NDF_PROBE_ROUND0=1 bash tools/_run3.sh 2>&1 | grep -v "^\"" | tail -3
python tools/s5_time.py 1.0 "" "CATCHHIP_NDF_PROBE_ROUND0=1" "CATCHHIP_FRONT_END_WORKERS=1" 2>&1 | tail -3
