export CATCHHIP_TEST_HOOKS=1
python -m pytest tests -m gpu -x -q -k "config5 or chains or design or minhash" 2>&1 | tail -2
python tools/s5_time.py 1.0 "" 2>&1 | tail -1
python tools/s5_time.py 1.0 "" 2>&1 | tail -1
