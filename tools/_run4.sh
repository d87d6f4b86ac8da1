export CATCHHIP_TEST_HOOKS=1
python -m pytest tests -m gpu -q 2>&1 | tail -8
