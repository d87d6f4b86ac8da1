export CATCHHIP_TEST_HOOKS=1
python tools/m2_timeline.py 2>&1 | head -5
CATCHHIP_BUILDERS=1 python tools/m2_timeline.py 2>&1 | head -5 | tail -4
CATCHHIP_LANE_FIXED_MBASES=0 CATCHHIP_BUILDERS=1 python tools/m2_timeline.py 2>&1 | head -5 | tail -4
CATCHHIP_LANE_FIXED_MBASES=25 python tools/m2_timeline.py 2>&1 | head -5 | tail -4
python -m pytest tests -m gpu -x -q -k "front_end or union or design or dropin or plugin or device_candidates" 2>&1 | tail -3
