export CATCHHIP_TEST_HOOKS=1
python -m pytest tests -m gpu -x -q -k "minhash or ndf or config5 or chains" 2>&1 | tail -3
export CATCHHIP_FRONT_END_WORKERS=1 CATCHHIP_PREFETCH_DEPTH=0
CATCHHIP_TIMING=2 python tools/s5_profile.py 1.0 once > gpurun_out/s5_trace_q.out 2> gpurun_out/s5_trace_q.err
CATCHHIP_MH_NO_WAVE64=1 CATCHHIP_TIMING=2 python tools/s5_profile.py 1.0 once > gpurun_out/s5_trace_q_old64.out 2> gpurun_out/s5_trace_q_old64.err
grep -c "lazy round" gpurun_out/s5_trace_q.err gpurun_out/s5_trace_q_old64.err
