FZ_TRIAL_LIB=build_tmp/libfz_trials.so python scripts/ab_lib_flag.py fz_igemm_trial_no_kg2 3 > $O/kg2_job_ab.txt 2>&1; tail -3 $O/kg2_job_ab.txt
