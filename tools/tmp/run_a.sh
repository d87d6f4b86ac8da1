export CATCHHIP_TEST_HOOKS=1
CATCHHIP_UNION_RAMP=off python tools/s5_time.py 1.0 "" 2>&1 | tail -1
python tools/s5_time.py 1.0 "" 2>&1 | tail -1
