FZ_TRIAL_LIB=build_tmp/libfz_trials.so python scripts/ab_lib_flag.py fz_igemm_trial_no_halo_split 2 24 > $O/halo_rule_24f_job_ab.txt 2>&1; tail -3 $O/halo_rule_24f_job_ab.txt
