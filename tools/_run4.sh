export CATCHHIP_TEST_HOOKS=1
bash tools/collect_units.sh r05 S5 1.0 1 > gpurun_out/units_S5.log 2>&1
bash tools/collect_units.sh r05 S3 1.0 3 > gpurun_out/units_S3.log 2>&1
bash tools/collect_profiles.sh r05 S4 > gpurun_out/profiles_S4.log 2>&1
tail -12 gpurun_out/units_S5.log | head -11
tail -12 gpurun_out/units_S3.log | head -11
tail -5 gpurun_out/profiles_S4.log | cut -c1-600
