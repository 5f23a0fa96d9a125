R=$GRAFT_REPO_ROOT; A=$R/$O; python scripts/job_breakdown.py > $A/job_breakdown.txt 2>&1; head -70 $A/job_breakdown.txt | cut -c1-170
