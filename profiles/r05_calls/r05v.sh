FZ_ISSUE_PLANS=1 timeout 400 python scripts/ab_bench.py fatezero_amd.issue ENABLED 2>&1 | tail -2 | tee -a $O/ab.txt
