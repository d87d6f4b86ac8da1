export CATCHHIP_TEST_HOOKS=1
python -m pytest tests -m gpu -x -q -k "cluster or config5 or signatures or designer" 2>&1 | tail -3
bash tools/_run3.sh 2>&1 | grep " ms " | head -6
python tools/s5_time.py 1.0 "" 2>&1 | tail -1
