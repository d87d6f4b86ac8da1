export CATCHHIP_TEST_HOOKS=1
python -m pytest tests -m gpu -x -q -k "minhash or ndf or config5 or chains or candidates or design" 2>&1 | tail -3
bash tools/ndf_kernel_trace.sh 2>&1 | grep " ms " | head -14
python tools/s5_time.py 1.0 "" 2>&1 | tail -1
