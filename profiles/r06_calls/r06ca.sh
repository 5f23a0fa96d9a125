python scripts/ab_two_libs_job.py build_tmp/libfz_before_ptab.so fatezero_amd/libfatezero_hip.so 3 > $O/ptab_job_ab.txt 2>&1; tail -2 $O/ptab_job_ab.txt
