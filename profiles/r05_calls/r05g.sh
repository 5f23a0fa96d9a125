timeout 600 python scripts/job_breakdown.py 4 > $O/breakdown.txt 2>&1; head -90 $O/breakdown.txt
