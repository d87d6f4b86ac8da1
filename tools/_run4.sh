export CATCHHIP_TEST_HOOKS=1
python -m pytest tests -m gpu -x -q -k "extension_replays" 2>&1 | tail -15
