python scripts/trials/plans_two_jobs_check.py 6 2>&1 | grep -v amdgpu.ids | tail -6
