python scripts/ff_chain_ab.py 4 8 16 24 32 > $O/ff_chain_ab_v2.txt 2>&1; tail -30 $O/ff_chain_ab_v2.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "feed_forward" > $O/gpu_tests_ff.log 2>&1; tail -3 $O/gpu_tests_ff.log
