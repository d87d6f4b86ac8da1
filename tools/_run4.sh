export CATCHHIP_TEST_HOOKS=1
python -m pytest tests -m gpu -x -q -k "minhash or ndf or config5 or chains" 2>&1 | tail -3
bash tools/_run3.sh 2>&1 | grep -v "^\"" | head -120
