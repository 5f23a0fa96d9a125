timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or conv or tconv or temporal" 2>&1 | tail -2
python scripts/ab_two_libs_gemm.py build_tmp/libfz_prev_igemm.so fatezero_amd/libfatezero_hip.so > $O/bias_lds_ab.txt 2>&1; cat $O/bias_lds_ab.txt
