# round-3 GPU call af: the complete -m gpu suite on the closing build, parity numbers printed (-s)
O=gpurun_out/r03af; mkdir -p $O
(timeout 640 python -m pytest tests -q -s -m gpu --durations=15) > $O/gpu_tests.log 2>&1
tail -22 $O/gpu_tests.log | cut -c1-200
