R=$GRAFT_REPO_ROOT
bash scripts/pmc_job.sh r04t_pmc_job 50 2>&1 | tail -2
