timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -2
FZ_TRIAL_LIB=build_tmp/libfz_trials.so python scripts/ab_lib_flag.py fz_igemm_trial_no_halo_split 3 > $O/halo_split_job_ab.txt 2>&1; tail -3 $O/halo_split_job_ab.txt
