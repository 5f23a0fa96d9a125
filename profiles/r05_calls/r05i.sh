(timeout 1700 python -m pytest tests -m gpu -q -s --durations=25) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
tail -40 $O/gpu_tests.log | cut -c1-220
