python scripts/ff_chain_ab.py > $O/ff_chain_ab.txt 2>&1; tail -20 $O/ff_chain_ab.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "feed_forward or gemm_lnout or geglu" > $O/gpu_tests_ff.log 2>&1; tail -5 $O/gpu_tests_ff.log
