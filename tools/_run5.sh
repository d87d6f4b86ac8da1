export CATCHHIP_TEST_HOOKS=1 S5_TIME_EVENTS=1
python tools/s5_time.py 1.0 "" 2>&1 | tail -62
